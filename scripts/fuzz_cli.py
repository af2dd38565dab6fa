"""Randomized sweep of the command-line boundary (dandd_amd/bin/dashing, dandd_amd/bin/fused/parallel; dandd_amd/csrc/dd_cli.c) against
the oracle, through ONE resident `dashing serve` (every command a client process, as DandD would start it): random FASTA texts
(kseq record shapes), register counts, canonical flag, k sets -- consecutive, scattered, unsorted, repeated --, one or several
FASTAs per `sketch`, the three sketch containers; per draw a k-batch through `parallel`, the same ks one `dashing sketch` at a time
(byte-identical files), a `union` over what was written and a multi-path `card`.  Every payload == oracle registers, every printed
cardinality == the oracle's estimate of those registers.      python scripts/fuzz_cli.py [N] [SEED]"""
import os, shutil, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dandd_amd import build
from dandd_amd.host.backend import read_sketch_file
from oracle import dd_oracle as orc

n_cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
build.build()
dashing, parallel = build.build_cli(), build.FUSED_PARALLEL
work = tempfile.mkdtemp(prefix="fuzz_cli_")
alph = [np.frombuffer(b"ACGT", np.uint8), np.frombuffer(b"ACGTacgtN", np.uint8), np.frombuffer(b"ACGTACGTACGTRYKMn-* 0>", np.uint8),
        np.frombuffer(b"ACGTACGTACGTACGTacgtN@+\r>", np.uint8)]


def text():
    parts = []
    for r in range(int(rng.integers(1, 5))):
        parts.append(b">rec %d\n" % r if rng.integers(0, 6) else b"@rec %d\n" % r)
        a = alph[int(rng.integers(0, 4))]
        total = int(rng.choice([0, 1, 50, 1000, 30000, 120000]) * rng.random()) + int(rng.integers(0, 3))
        width = int(rng.choice([1, 60, 61, 80, 10 ** 9]))
        seq = rng.choice(a, size=total).tobytes()
        parts.append(b"\n".join(seq[i:i + width] for i in range(0, len(seq), width)) + (b"\n" if rng.integers(0, 2) else b""))
    return b"".join(parts)


def run(exe, args, env, cwd):
    r = subprocess.run([exe] + args, env=env, capture_output=True, text=True, timeout=300, cwd=cwd)
    if r.returncode != 0:
        raise SystemExit(f"seed {seed}: {exe} {args} -> {r.returncode}\n{r.stdout[-800:]}{r.stderr[-800:]}")
    return r.stdout


sock = os.path.join(work, "d.sock")
base_env = dict(os.environ)
for v in ("DANDD_DASHING_SERVER", "DANDD_SKETCH_FORMAT"):
    base_env.pop(v, None)
servers = {}


def server(fmt):
    """one resident process per sketch container (the format is the SERVER's environment: INTEGRATION.md section 0)"""
    if fmt not in servers:
        s = os.path.join(work, f"{fmt}.sock")
        env = dict(base_env) if fmt == "native" else dict(base_env, DANDD_SKETCH_FORMAT=fmt)
        p = subprocess.Popen([dashing, "serve", "--socket", s], stdout=subprocess.PIPE, text=True, env=env)
        assert "listening" in p.stdout.readline()
        servers[fmt] = (p, dict(base_env, DANDD_DASHING_SERVER=s, DANDD_SERVER_REQUIRED="1"))
    return servers[fmt][1]


t0 = time.time()
checked = 0
try:
    for it in range(n_cfg):
        d = os.path.join(work, f"c{it}")
        os.makedirs(d)
        p = int(rng.choice([4, 8, 10, 12, 14, 16, 17, 18, 20]))
        canon = bool(rng.integers(0, 2))
        fmt = str(rng.choice(["native", "native", "dashing", "dashing-plain"]))
        cenv = server(fmt)
        nf = int(rng.choice([1, 1, 1, 2, 3]))
        fastas = []
        for f in range(nf):
            path = os.path.join(d, f"s{f}.fa")
            with open(path, "wb") as fh:
                fh.write(text())
            fastas.append(path)
        # the k set: a run, scattered values, unsorted, sometimes with a repeat
        kmax = 32 if rng.integers(0, 3) else 64
        ks = sorted(set(int(x) for x in rng.integers(1, kmax + 1, size=int(rng.integers(1, 7)))))
        if rng.integers(0, 2):
            k0 = int(rng.integers(1, kmax - 3))
            ks = sorted(set(ks + list(range(k0, k0 + int(rng.integers(2, 5))))))
        order = [str(k) for k in rng.permutation(ks)]
        if rng.integers(0, 5) == 0:
            order.append(order[0])
        for k in ks:
            os.makedirs(os.path.join(d, "a", f"k{k}"))
            os.makedirs(os.path.join(d, "b", f"k{k}"))
        flags = ("" if canon else "--no-canon ") + (f"-k{{}} -S {p}" if rng.integers(0, 2) else f"-k {{}} -S{p}")
        # (a) the k-batch as DandD's shell hands it over; (b) the same ks one `dashing sketch` at a time
        run(parallel, ["-j", "95%", f" dashing sketch {flags} --prefix {d}/a/k{{}} {' '.join(fastas)} ", ":::"] + order, cenv, d)
        for k in ks:
            run(dashing, ["sketch"] + ([] if canon else ["--no-canon"]) + [f"-k{k}", "-S", str(p), "--prefix", f"{d}/b/k{k}"] + fastas, cenv, d)
        regs = {}
        for f in fastas:
            buf = np.fromfile(f, dtype=np.uint8)
            want = orc.sketch_sweep(buf, ks[0], ks[-1], p, canon)
            for k in ks:
                name = f"k{k}/{os.path.basename(f)}.w.{k}.spacing.{p}.hll"
                a, b = os.path.join(d, "a", name), os.path.join(d, "b", name)
                got, lp, kk, _c = read_sketch_file(a)
                if not (lp == p and kk == k and np.array_equal(got, want[k - ks[0]])):
                    raise SystemExit(f"seed {seed} draw {it}: {a} differs from the oracle (p={p} canon={canon} k={k} fmt={fmt})")
                if fmt != "dashing" and open(a, "rb").read() != open(b, "rb").read():       # (gzip'd containers carry no name or time, but compare payloads anyway)
                    raise SystemExit(f"seed {seed} draw {it}: {a} and {b} differ")
                if fmt == "dashing" and not np.array_equal(read_sketch_file(b)[0], got):
                    raise SystemExit(f"seed {seed} draw {it}: {a} and {b} differ")
                regs[a] = got
                checked += 1
        # a union per k over the FASTAs' sketches (a `parallel` k-batch of `dashing union`, as DandD wraps those too), then one `card` for all
        if nf > 1:
            ins = " ".join(f"{d}/a/k{{}}/{os.path.basename(f)}.w.{{}}.spacing.{p}.hll" for f in fastas)
            run(parallel, ["-j", "95%", f" dashing union -z -o {d}/a/k{{}}/u_{p}n{nf}k{{}}{'' if canon else 'nc'}.hll {ins} ", ":::"] + [str(k) for k in ks], cenv, d)
            for k in ks:
                u = f"{d}/a/k{k}/u_{p}n{nf}k{k}{'' if canon else 'nc'}.hll"
                got, lp, kk, cc = read_sketch_file(u)
                want = np.maximum.reduce([regs[f"{d}/a/k{k}/{os.path.basename(f)}.w.{k}.spacing.{p}.hll"] for f in fastas])
                if not (np.array_equal(got, want) and lp == p and kk == k and cc == canon):
                    raise SystemExit(f"seed {seed} draw {it}: union {u} is not the byte max of its inputs")
                regs[u] = got
                checked += 1
        paths = list(regs)
        out = run(dashing, ["card", "--presketched"] + paths, cenv, d)
        lines = out.splitlines()
        assert lines[0] == "#Path\tSize (est.)" and len(lines) == 1 + len(paths)
        for line, path in zip(lines[1:], paths):
            got_path, v = line.split("\t")
            if got_path != path or float(v) != orc.card(regs[path], p):
                raise SystemExit(f"seed {seed} draw {it}: card of {path}: {v} against {orc.card(regs[path], p)!r}")
        shutil.rmtree(d)
    print(f"{n_cfg} random command-line draws ({checked} sketch files through `parallel` and `dashing`, three containers): payloads == oracle, "
          f"cards == oracle, fused == one by one, in {time.time() - t0:.1f} s")
finally:
    for p_, cenv in servers.values():
        subprocess.run([dashing, "shutdown"], env=cenv, capture_output=True)
        try:
            p_.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p_.kill()
    shutil.rmtree(work, ignore_errors=True)
