"""DandD's k-batch (`parallel -j 95% 'dashing sketch -k{} ...' ::: k...`, /root/reference/lib/huffman_dandd.py:214-218) over ONE genome:
what the call costs as K `dashing` processes the way GNU parallel starts them (15 at a time; fresh processes, or clients of a
resident `dashing serve`) and as the ONE fused sweep dandd_amd/bin/fused/parallel makes of it.
    python scripts/kbatch_probe.py [MBP=50] [LOG2M=14] [KMIN=4] [KMAX=40]"""
import os, shutil, subprocess, sys, tempfile, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dandd_amd import build
from oracle import dd_oracle as orc

mbp, p, kmin, kmax = (int(sys.argv[i]) if len(sys.argv) > i else d for i, d in ((1, 50), (2, 14), (3, 4), (4, 40)))
build.build()
dashing, parallel = build.build_cli(), build.FUSED_PARALLEL
work = tempfile.mkdtemp(prefix="kbatch_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
fasta = os.path.join(work, "g.fasta")
orc.synth_fasta(77, 0, mbp * 1_000_000, 5).tofile(fasta)
ks = list(range(kmin, kmax + 1))
env = dict(os.environ)
env.pop("DANDD_DASHING_SERVER", None)


def fresh(tag):
    d = os.path.join(work, tag)
    for k in ks:
        os.makedirs(os.path.join(d, f"k{k}"))
    return d


def one_by_one(tag, e):
    d = fresh(tag)
    def run(k):
        return subprocess.run([dashing, "sketch", f"-k{k}", "-S", str(p), "--prefix", os.path.join(d, f"k{k}"), fasta], env=e, capture_output=True).returncode
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=max(1, int(0.95 * (os.cpu_count() or 1)))) as pool:
        assert not any(pool.map(run, ks))
    return time.perf_counter() - t0, d


def fused(tag, e):
    d = fresh(tag)
    t0 = time.perf_counter()
    r = subprocess.run([parallel, "-j", "95%", f" dashing sketch  -k{{}} -S {p} --prefix {d}/k{{}} {fasta} ", ":::"] + [str(k) for k in ks], env=e, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return time.perf_counter() - t0, d


def same(a, b):
    for k in ks:
        n = f"k{k}/g.fasta.w.{k}.spacing.{p}.hll"
        assert open(os.path.join(a, n), "rb").read() == open(os.path.join(b, n), "rb").read(), n


try:
    t_proc, d0 = one_by_one("procs", env)
    t_fused, d1 = fused("fused", env)
    same(d0, d1)
    sock = os.path.join(work, "d.sock")
    srv = subprocess.Popen([dashing, "serve", "--socket", sock], stdout=subprocess.PIPE, text=True, env=env)
    assert "listening" in srv.stdout.readline()
    cenv = dict(env, DANDD_DASHING_SERVER=sock, DANDD_SERVER_REQUIRED="1")
    one_by_one("warm", cenv)                      # (the server's first commands bring its context up)
    t_srv, d2 = one_by_one("srv", cenv)
    fused("warm2", cenv)
    t_srv_fused, d3 = fused("srv_fused", cenv)
    same(d0, d2), same(d0, d3)
    # the `card` DandD asks for after a k-batch: all K sketches in one command (lib/sketch_classes.py:306-316), through a resident
    # server of the build under test (KBATCH_OTHER_DASHING: another build's executable, for an A/B on the same files)
    sketches = [os.path.join(d3, f"k{k}", f"g.fasta.w.{k}.spacing.{p}.hll") for k in ks]
    t_card = []
    for n, exe in enumerate([dashing] + ([os.environ["KBATCH_OTHER_DASHING"]] if os.environ.get("KBATCH_OTHER_DASHING") else [])):
        s2 = os.path.join(work, f"card{n}.sock")
        srv2 = subprocess.Popen([exe, "serve", "--socket", s2], stdout=subprocess.PIPE, text=True, env=env)
        assert "listening" in srv2.stdout.readline()
        e2 = dict(env, DANDD_DASHING_SERVER=s2, DANDD_SERVER_REQUIRED="1")
        best = 1e9
        for _ in range(6):
            t0 = time.perf_counter()
            r = subprocess.run([exe, "card", "--presketched"] + sketches, env=e2, capture_output=True, text=True)
            best = min(best, time.perf_counter() - t0)
            assert r.returncode == 0 and len(r.stdout.splitlines()) == len(ks) + 1, r.stderr
        subprocess.run([exe, "shutdown"], env=e2, capture_output=True)
        srv2.wait(timeout=60)
        t_card.append((exe, best, r.stdout))
    assert all(o == t_card[0][2] for _, _, o in t_card)
    subprocess.run([dashing, "shutdown"], env=cenv, capture_output=True)
    srv.wait(timeout=60)
    K = len(ks)
    print(f"one {mbp} Mbp genome, k {kmin}-{kmax} (K = {K}), -S {p}; the K files are byte-identical in all four")
    print(f"  K fresh `dashing sketch` processes, {max(1, int(0.95 * (os.cpu_count() or 1)))} at a time : {t_proc:7.3f} s")
    print(f"  the same K as clients of `dashing serve`                     : {t_srv:7.3f} s")
    print(f"  ONE fused `parallel` process                                  : {t_fused:7.3f} s   ({mbp / 1e3 / t_fused:.2f} Gbp/s of sequence, all K ks, end to end)")
    print(f"  ONE fused `parallel` through `dashing serve`                  : {t_srv_fused:7.3f} s   ({mbp / 1e3 / t_srv_fused:.2f} Gbp/s of sequence, all K ks, end to end)")
    for exe, best, _ in t_card:
        print(f"  `card --presketched` of the K sketches, client {os.path.relpath(exe, ROOT):32s}: {best:7.3f} s (best of 6, through its own resident server)")
finally:
    shutil.rmtree(work, ignore_errors=True)
