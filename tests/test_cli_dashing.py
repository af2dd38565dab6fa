"""`dandd_amd/bin/dashing` (dandd_amd/csrc/dd_cli.c): the reference's real plugin API -- the `dashing sketch | union | card` argv +
files + stdout contract of /root/reference/lib/sketch_classes.py:306-321,351-373 and lib/huffman_dandd.py:214-218 -- as a plain C
program over the C ABI.

GPU: the command trace the UNMODIFIED reference issued (tests/golden/ref_trace_hll.json, captured by tests/golden/make_golden.py
with oracle-backed shims on PATH) is replayed line by line through the product's executable; every `card` answer must be the
cardinality the reference cached, every `.hll` payload the oracle's registers.
CPU: the executable builds, refuses to run without a GPU (no fallback), rejects bad command lines, and its resident form answers."""
import hashlib
import json
import os
import shutil
import subprocess
import time

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
NATIVE_HEAD = 12


@pytest.fixture(scope="module")
def dashing():
    from dandd_amd import build
    build.build()
    return build.build_cli()


@pytest.fixture(scope="module")
def parallel(dashing):
    from dandd_amd import build
    assert os.path.exists(build.FUSED_PARALLEL)
    return build.FUSED_PARALLEL


def _trace():
    with open(os.path.join(GOLD, "ref_trace_hll.json")) as f:
        return json.load(f)


def _start_server(dashing, sock, env):
    srv = subprocess.Popen([dashing, "serve", "--socket", sock, "--idle-exit", "600"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    line = srv.stdout.readline()
    assert "listening" in line, (line, srv.stderr.read() if srv.poll() is not None else "")
    return srv


def test_trace_fixture_is_the_reference_grammar():
    """the fixture holds only the three command shapes SURVEY.md section 8(b) lists, and a cached cardinality for every sketch"""
    t = _trace()
    assert len(t["commands"]) > 700
    printed = {}
    for c in t["commands"]:
        a = c["argv"]
        assert a[0] == "dashing" and a[1] in ("sketch", "union", "card")
        if a[1] == "sketch":
            rest = [x for x in a[2:] if x != "--no-canon"]
            assert rest[0].startswith("-k") and rest[1] == "-S" and rest[3] == "--prefix" and len(rest) == 6
            assert os.path.basename(c["out"]) == f"{os.path.basename(rest[5])}.w.{rest[0][2:]}.spacing.{rest[2]}.hll"
        elif a[1] == "union":
            assert a[2:4] == ["-z", "-o"] and len(a) >= 6 and c["out"] == a[4]
        else:
            assert a[2] == "--presketched" and len(a) == 4      # (always ONE path: SURVEY.md section 8a, row a4)
            printed.update(c["cards"])
    cached = {}
    for table in t["cardinality_caches"].values():
        cached.update(table)
    assert cached and all(printed.get(p) == v for p, v in cached.items())
    # the k-batches: every `sketch` and `union` reached Dashing through ONE `parallel -j 95% '<template>' ::: k ...` per node
    # (lib/huffman_dandd.py:214-218); the fixture keeps those calls as the reference's shell handed them over
    covered = set()
    for b in t["parallel_calls"]:
        assert b["argv"][:3] == ["parallel", "-j", "95%"] and b["argv"][4] == ":::" and b["n"] == len(b["argv"]) - 5
        for j, v in enumerate(b["argv"][5:]):
            assert t["commands"][b["first"] + j]["argv"] == b["argv"][3].replace("{}", v).split()
            covered.add(b["first"] + j)
    assert covered == {i for i, c in enumerate(t["commands"]) if c["argv"][1] in ("sketch", "union")}


def test_cli_without_gpu_fails_loudly(dashing, tmp_path):
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    env.pop("DANDD_DASHING_SERVER", None)
    r = subprocess.run([dashing, "sketch", "-k9", "-S", "12", "--prefix", str(tmp_path), os.path.join(GOLD, "fasta", "g0.fasta")],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "no CPU path" in r.stderr, (r.returncode, r.stderr)
    assert not os.listdir(tmp_path)                               # nothing half-written takes a sketch's name
    # the thread flag the reference has commented out (lib/sketch_classes.py:313,361,370) parses, glued or apart: the call gets as far as the GPU
    for threads in (["-p10"], ["-p", "10"]):
        r = subprocess.run([dashing, "sketch"] + threads + ["-k9", "-S", "12", "--prefix", str(tmp_path), os.path.join(GOLD, "fasta", "g0.fasta")],
                           env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "no CPU path" in r.stderr, (threads, r.returncode, r.stderr)
    for bad in (["frobnicate"], ["sketch", "-k9", "--prefix", str(tmp_path)], ["sketch", "-k99", "-S", "12", "x.fa"], ["union", "-o"],
                ["card", "--frob", "x"], []):
        r = subprocess.run([dashing] + bad, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 64 and r.stderr, (bad, r.returncode, r.stderr)
    r = subprocess.run([dashing, "card", "--presketched", os.path.join(GOLD, "fasta", "g0.fasta")], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "neither a dandd_amd nor a Dashing sketch file" in r.stderr
    assert r.stdout == "#Path\tSize (est.)\n"


def test_parallel_name_takes_only_the_k_batch(parallel, dashing, tmp_path):
    """dandd_amd/bin/fused/parallel: DandD's k-batch is run here (without a GPU: fails loudly, writes nothing); anything else goes
    to the next `parallel` on PATH with its argv untouched, or -- none there -- value by value through /bin/sh with GNU parallel's
    exit status (the number of failed jobs)."""
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    env.pop("DANDD_DASHING_SERVER", None)
    fasta = os.path.join(GOLD, "fasta", "g0.fasta")
    batch = ["-j", "95%", f" dashing sketch  -k{{}} -S 12 --prefix {tmp_path}/k{{}} {fasta} ", ":::", "9", "10", "11"]
    for exe in ([parallel], [dashing, "parallel"]):
        r = subprocess.run(exe + batch, env=env, capture_output=True, text=True, timeout=120)
        assert r.returncode == 1 and "no CPU path" in r.stderr and not os.listdir(tmp_path), (exe, r.returncode, r.stderr)
    # no other `parallel` on PATH: the simplest GNU-parallel form, one job at a time
    bare = dict(env, PATH=os.path.dirname(parallel) + ":/usr/bin:/bin")
    if not any(os.path.exists(os.path.join(d, "parallel")) for d in ("/usr/bin", "/bin")):
        r = subprocess.run([parallel, "-j", "2", "echo a{}b; test {} != 2", ":::", "1", "2", "3"], env=bare, capture_output=True, text=True, timeout=120)
        assert r.stdout == "a1b\na2b\na3b\n" and r.returncode == 1, (r.returncode, r.stdout, r.stderr)
        r = subprocess.run([parallel, "--bar", "echo {}", ":::", "1"], env=bare, capture_output=True, text=True, timeout=120)
        assert r.returncode == 255 and "no other `parallel`" in r.stderr
    # another `parallel` further down PATH: it gets everything that is not the k-batch, argv as given
    other = tmp_path / "gnu"
    other.mkdir()
    (other / "parallel").write_text("#!/bin/sh\necho REAL \"$@\"\nexit 7\n")
    os.chmod(other / "parallel", 0o755)
    withreal = dict(env, PATH=os.path.dirname(parallel) + ":" + str(other) + ":/usr/bin:/bin")
    for args in (["-j", "4", "gzip {}", ":::", "a.txt", "b.txt"], ["--bar", "-j", "95%", " dashing sketch -k{} -S 12 x.fa ", ":::", "9"],
                 ["-j", "95%", " dashing sketch -k{} -S 12 x.fa | tee log ", ":::", "9"], ["-j", "95%", " dashing card --presketched {} ", ":::", "a.hll"]):
        r = subprocess.run([parallel] + args, env=withreal, capture_output=True, text=True, timeout=120)
        assert r.returncode == 7 and r.stdout == "REAL " + " ".join(args) + "\n", (args, r.returncode, r.stdout, r.stderr)
    r = subprocess.run([parallel] + batch, env=withreal, capture_output=True, text=True, timeout=120)      # (the k-batch stays here)
    assert r.returncode == 1 and "no CPU path" in r.stderr and "REAL" not in r.stdout


def test_resident_form_answers_and_survives_bad_clients(dashing, tmp_path):
    """`dashing serve`: ping, a command that fails (no GPU here) with its message sent back, a client that hangs up mid-request, shutdown"""
    import socket
    sock = str(tmp_path / "d.sock")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    env.pop("DANDD_DASHING_SERVER", None)
    srv = _start_server(dashing, sock, env)
    try:
        cenv = dict(env, DANDD_DASHING_SERVER=sock, DANDD_SERVER_REQUIRED="1")
        assert oct(os.stat(sock).st_mode & 0o777) == "0o600"
        r = subprocess.run([dashing, "ping"], env=cenv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and int(r.stdout) == srv.pid
        for junk in (b"", b"\x02\x00", b"\xff\xff\xff\xff", b"\x01\x00\x00\x00\x10\x00\x00\x00abc"):
            s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            s.connect(sock)
            s.sendall(junk)
            s.close()
        # clients that send a whole request and hang up before the reply is written (a DandD killed mid-command): the write fails,
        # the server stays (it died of SIGPIPE before round 6's review of its own code)
        import struct
        blob = lambda b: struct.pack("<I", len(b)) + b
        for _ in range(300):
            s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            s.connect(sock)
            s.sendall(struct.pack("<I", 3) + blob(b"card") + blob(b"--presketched") + blob(b"no/such.hll") + blob(str(tmp_path).encode()))
            s.close()
        assert srv.poll() is None
        r = subprocess.run([dashing, "card", "--presketched", "no/such.hll"], env=cenv, capture_output=True, text=True, timeout=60, cwd=str(tmp_path))
        assert r.returncode == 1 and "no/such.hll" in r.stderr and r.stdout == "#Path\tSize (est.)\n"
        r = subprocess.run([dashing, "sketch", "-k9", "-S", "12", "--prefix", ".", os.path.join(GOLD, "fasta", "g0.fasta")], env=cenv,
                           capture_output=True, text=True, timeout=120, cwd=str(tmp_path))
        assert r.returncode == 1 and "no CPU path" in r.stderr
        r = subprocess.run([dashing, "shutdown"], env=cenv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0
        assert srv.wait(timeout=30) == 0 and not os.path.exists(sock)
        r = subprocess.run([dashing, "ping"], env=cenv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 111 and "no server" in r.stderr
    finally:
        if srv.poll() is None:
            srv.kill()


def _replay(dashing, work, commands, env, orc, check_oracle, batches=(), parallel=None):
    """run every traced line in `work`; -> number of processes started.  With `batches` (the fixture's parallel_calls) the lines
    that went through GNU parallel are run as the ONE `parallel` call the reference made, by `parallel` (the product's
    dandd_amd/bin/fused/parallel), and checked output by output as if each had run alone."""
    regs_of = {}                                                   # output path -> registers, for checking unions against their inputs
    starts = {b["first"]: b for b in batches}
    started = 0

    def prepare(argv):
        if argv[1] == "sketch":
            os.makedirs(argv[argv.index("--prefix") + 1], exist_ok=True)     # (the reference makes the per-k directories itself: lib/huffman_dandd.py:171-174)
        elif argv[1] == "union":
            os.makedirs(os.path.dirname(argv[argv.index("-o") + 1]), exist_ok=True)

    def check_output(c, argv):
        out = c["out"].replace("@W@", work)
        raw = np.fromfile(out, dtype=np.uint8)
        assert raw[:8].tobytes() == b"DDHLL\x01\x00\x00"
        regs = raw[NATIVE_HEAD:]
        assert hashlib.sha256(regs.tobytes()).hexdigest() == c["regs_sha256"], argv
        regs_of[out] = regs
        if check_oracle:
            if argv[1] == "sketch":
                rest = [x for x in argv[2:] if x != "--no-canon"]
                k, p, fasta = int(rest[0][2:]), int(rest[2]), rest[5]
                want = orc.sketch(np.fromfile(fasta, dtype=np.uint8), k, p, "--no-canon" not in argv)
                assert raw[8] == p and raw[9] == k and raw[10] == int("--no-canon" not in argv)
            else:
                ins = argv[argv.index("-o") + 2:]
                want = np.maximum.reduce([regs_of[i] if i in regs_of else np.fromfile(i, dtype=np.uint8)[NATIVE_HEAD:] for i in ins])
            assert np.array_equal(regs, want), argv

    i = 0
    while i < len(commands):
        if i in starts:
            b = starts[i]
            covered = commands[i:i + b["n"]]
            argvs = [[x.replace("@W@", work) for x in c["argv"]] for c in covered]
            for argv in argvs:
                prepare(argv)
            r = subprocess.run([parallel] + [x.replace("@W@", work) for x in b["argv"][1:]], env=env, capture_output=True, text=True, timeout=300, cwd=work)
            assert r.returncode == 0, (b["argv"], r.returncode, r.stdout, r.stderr)
            started += 1
            for c, argv in zip(covered, argvs):
                check_output(c, argv)
            i += b["n"]
            continue
        c = commands[i]
        i += 1
        argv = [x.replace("@W@", work) for x in c["argv"]]
        prepare(argv)
        r = subprocess.run([dashing] + argv[1:], env=env, capture_output=True, text=True, timeout=300, cwd=work)
        started += 1
        assert r.returncode == 0, (argv, r.returncode, r.stdout, r.stderr)
        if argv[1] == "card":
            lines = r.stdout.splitlines()
            assert lines[0] == "#Path\tSize (est.)" and len(lines) == 1 + len(c["cards"])
            for line, (path, want) in zip(lines[1:], c["cards"].items()):
                got_path, got = line.split("\t")
                assert got_path == path.replace("@W@", work)
                assert float(got) == float(want), (argv, got, want)      # the double the reference put in its cardkey (lib/sketch_classes.py:318-321)
            continue
        check_output(c, argv)
    return started


@pytest.mark.gpu
def test_reference_command_trace_replayed_through_the_product(dashing, tmp_path, orc, torch_cuda):
    """Every command line the unmodified reference issued over eleven CLI scenarios (742 of them), through the resident form
    (one process keeps the GPU contexts; the traced argv is what a client process sends), in a directory laid out as the
    reference laid it out.  Then the reference's own cardinality caches, asked for again in one multi-path `card`."""
    t = _trace()
    work = str(tmp_path / "w")
    shutil.copytree(os.path.join(GOLD, "fasta"), os.path.join(work, "data"))
    sock = str(tmp_path / "d.sock")
    env = dict(os.environ)
    env.pop("DANDD_DASHING_SERVER", None)
    env.pop("DANDD_SKETCH_FORMAT", None)
    srv = _start_server(dashing, sock, env)
    try:
        cenv = dict(env, DANDD_DASHING_SERVER=sock, DANDD_SERVER_REQUIRED="1")
        t0 = time.perf_counter()
        n = _replay(dashing, work, t["commands"], cenv, orc, check_oracle=True)
        dt = time.perf_counter() - t0
        print(f"replayed {n} traced commands through `dashing serve` in {dt:.1f} s ({1e3 * dt / n:.1f} ms per command, checks included)")
        for cache, table in t["cardinality_caches"].items():
            paths = [p.replace("@W@", work) for p in table]
            r = subprocess.run([dashing, "card", "--presketched"] + paths, env=cenv, capture_output=True, text=True, timeout=300, cwd=work)
            assert r.returncode == 0, r.stderr
            got = dict(line.split("\t") for line in r.stdout.splitlines()[1:])
            assert {p: float(v) for p, v in got.items()} == {p.replace("@W@", work): float(v) for p, v in table.items()}, cache
        r = subprocess.run([dashing, "shutdown"], env=cenv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and int(r.stdout) >= n
        assert srv.wait(timeout=60) == 0
    finally:
        if srv.poll() is None:
            srv.kill()


@pytest.mark.gpu
def test_one_process_per_command_like_the_reference(dashing, tmp_path, orc, torch_cuda):
    """The literal contract: no server, a fresh process (and a fresh GPU context) per traced line -- the first scenario's hill-climb
    on the first leaf and what follows it (sketch x3 under `parallel`, card x3, ...), in both sketch containers."""
    t = _trace()
    env = dict(os.environ)
    env.pop("DANDD_DASHING_SERVER", None)
    env.pop("DANDD_SKETCH_FORMAT", None)
    work = str(tmp_path / "w")
    shutil.copytree(os.path.join(GOLD, "fasta"), os.path.join(work, "data"))
    first = t["commands"][:24]
    assert {c["argv"][1] for c in first} >= {"sketch", "card"}
    _replay(dashing, work, first, env, orc, check_oracle=True)
    # a union of the leaves sketched above, written in Dashing's container as recalled (gzip), read back by `card` and by the store
    sk = [c["out"].replace("@W@", work) for c in first if c["argv"][1] == "sketch"]
    k = int(first[0]["argv"][2][2:])
    same_k = [s for s in sk if f".w.{k}.spacing." in s]
    out = os.path.join(work, f"u_12n{len(same_k)}k{k}.hll")
    denv = dict(env, DANDD_SKETCH_FORMAT="dashing")
    r = subprocess.run([dashing, "union", "-z", "-o", out] + same_k, env=denv, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.read(2) == b"\x1f\x8b"
    from dandd_amd.host.backend import read_sketch_file
    regs, log2m, kk, canon = read_sketch_file(out)
    want = np.maximum.reduce([np.fromfile(s, dtype=np.uint8)[NATIVE_HEAD:] for s in same_k])
    assert np.array_equal(regs, want) and (log2m, kk, canon) == (12, k, True)
    r = subprocess.run([dashing, "card", "--presketched", out], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and float(r.stdout.splitlines()[1].split("\t")[1]) == orc.card(want, 12)
    # several FASTAs in one `sketch` (Dashing takes them; the reference never does): one file each, through the ingestion pipeline,
    # non-canonical, k and S as separate arguments; a multi-path `card` over them in Dashing's plain layout
    many = os.path.join(work, "many")
    os.makedirs(many)
    fastas = [os.path.join(work, "data", f"g{g}.fasta") for g in (0, 2, 4)]
    penv = dict(env, DANDD_SKETCH_FORMAT="dashing-plain")
    r = subprocess.run([dashing, "sketch", "--no-canon", "-k", "13", "-S", "10", "--prefix", many] + fastas, env=penv, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    outs = [os.path.join(many, os.path.basename(f) + ".w.13.spacing.10.hll") for f in fastas]
    for f, o in zip(fastas, outs):
        regs, log2m, kk, _canon = read_sketch_file(o)
        assert (log2m, kk) == (10, 13) and np.array_equal(regs, orc.sketch(np.fromfile(f, dtype=np.uint8), 13, 10, False))
    r = subprocess.run([dashing, "card", "--presketched"] + outs, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and len(r.stdout.splitlines()) == 4
    for line, f in zip(r.stdout.splitlines()[1:], fastas):
        assert float(line.split("\t")[1]) == orc.card(orc.sketch(np.fromfile(f, dtype=np.uint8), 13, 10, False), 10)


@pytest.mark.gpu
def test_cli_at_baseline_genome_size_and_default_registers(dashing, tmp_path, orc, engine_factory):
    """Three 50 Mbp genomes (BASELINE cfg 2's genome size) at DandD's default `-S 20` through the executable, one process per
    command: each payload == the library called directly == (for one genome) the oracle; the union's payload is the byte-wise
    max of its inputs; `card` prints the double the library's estimator returns, and it is within 3 sigma (1.04 / sqrt(2^20))
    of the distinct canonical 21-mers of a uniform random genome (~ its length less its N blocks)."""
    env = dict(os.environ)
    env.pop("DANDD_DASHING_SERVER", None)
    env.pop("DANDD_SKETCH_FORMAT", None)
    k, p, nb = 21, 20, 50_000_000
    data = tmp_path / "data"
    data.mkdir()
    fastas = []
    for g in range(3):
        f = data / f"big{g}.fasta"
        f.write_bytes(orc.synth_fasta(424242, g, nb, 7).tobytes())
        fastas.append(str(f))
    eng = engine_factory(p, True)
    direct = eng.sketch_files(fastas, k, k)                            # [3][1][2^20]
    outs = []
    for g, f in enumerate(fastas):
        r = subprocess.run([dashing, "sketch", f"-k{k}", "-S", str(p), "--prefix", str(tmp_path), f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr
        out = str(tmp_path / f"big{g}.fasta.w.{k}.spacing.{p}.hll")
        regs = np.fromfile(out, dtype=np.uint8)[NATIVE_HEAD:]
        assert regs.size == 1 << p and np.array_equal(regs, direct[g, 0])
        outs.append(out)
    assert np.array_equal(direct[0, 0], orc.sketch(np.fromfile(fastas[0], dtype=np.uint8), k, p, True))
    u = str(tmp_path / "u.hll")
    r = subprocess.run([dashing, "union", "-z", "-o", u] + outs, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    merged = np.fromfile(u, dtype=np.uint8)[NATIVE_HEAD:]
    assert np.array_equal(merged, np.maximum.reduce([direct[g, 0] for g in range(3)]))
    r = subprocess.run([dashing, "card", "--presketched", u] + outs, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = [float(line.split("\t")[1]) for line in r.stdout.splitlines()[1:]]
    assert got == [float(eng.card(merged))] + [float(eng.card(direct[g, 0])) for g in range(3)]
    sigma = 1.04 / (1 << p) ** 0.5
    for est in got[1:]:                                             # (the three genomes are 1 % apart: their union is not 3 x nb)
        assert abs(est / nb - 1.0) < 3 * sigma + 3e-3, est          # (3e-3: the 0.1 % of 100-base blocks that are N take ~120 windows each)
    assert max(got[1:]) < got[0] < sum(got[1:])


@pytest.mark.gpu
def test_reference_k_batches_run_fused(dashing, parallel, tmp_path, orc, torch_cuda):
    """The reference's k-batch call site itself (lib/huffman_dandd.py:214-218,233): each of the 163 `parallel -j 95% '<dashing ...
    {} ...>' ::: k ...` calls the unmodified reference made goes, argv as its shell handed it over, to the product's `parallel`
    (one process, ONE fused sweep over the node's FASTA for all its ks / its unions k by k), the 371 `card` lines to `dashing`.
    Every file a batch leaves is checked as if its K processes had run: digest == the fixture's, registers == the oracle's.
    Through the resident form, then the first batches as fresh processes."""
    t = _trace()
    work = str(tmp_path / "w")
    shutil.copytree(os.path.join(GOLD, "fasta"), os.path.join(work, "data"))
    sock = str(tmp_path / "d.sock")
    env = dict(os.environ)
    env.pop("DANDD_DASHING_SERVER", None)
    env.pop("DANDD_SKETCH_FORMAT", None)
    srv = _start_server(dashing, sock, env)
    try:
        cenv = dict(env, DANDD_DASHING_SERVER=sock, DANDD_SERVER_REQUIRED="1")
        t0 = time.perf_counter()
        n = _replay(dashing, work, t["commands"], cenv, orc, check_oracle=True, batches=t["parallel_calls"], parallel=parallel)
        dt = time.perf_counter() - t0
        assert n == len(t["parallel_calls"]) + sum(c["argv"][1] == "card" for c in t["commands"])
        print(f"replayed {len(t['commands'])} traced commands as {n} processes ({len(t['parallel_calls'])} fused k-batches) in {dt:.1f} s")
        r = subprocess.run([dashing, "shutdown"], env=cenv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and int(r.stdout) == n
        assert srv.wait(timeout=60) == 0
    finally:
        if srv.poll() is None:
            srv.kill()
    work2 = str(tmp_path / "w2")
    shutil.copytree(os.path.join(GOLD, "fasta"), os.path.join(work2, "data"))
    upto = t["parallel_calls"][5]["first"]
    _replay(dashing, work2, t["commands"][:upto], env, orc, check_oracle=True, batches=t["parallel_calls"][:5], parallel=parallel)
    # a batch whose ks are not consecutive and come unsorted, non-canonical: two sweeps, five files
    many = os.path.join(work2, "many")
    fasta = os.path.join(work2, "data", "g3.fasta")
    for k in (7, 8, 9, 21, 22):
        os.makedirs(os.path.join(many, f"k{k}"))
    r = subprocess.run([parallel, "-j", "95%", f" dashing sketch --no-canon -k{{}} -S 11 --prefix {many}/k{{}} {fasta} ", ":::", "22", "7", "9", "8", "21"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for k in (7, 8, 9, 21, 22):
        raw = np.fromfile(os.path.join(many, f"k{k}", f"g3.fasta.w.{k}.spacing.11.hll"), dtype=np.uint8)
        assert raw[8] == 11 and raw[9] == k and raw[10] == 0
        assert np.array_equal(raw[NATIVE_HEAD:], orc.sketch(np.fromfile(fasta, dtype=np.uint8), k, 11, False))


@pytest.mark.gpu
def test_forty_clients_at_once_like_gnu_parallel(dashing, tmp_path, orc, torch_cuda):
    """What GNU parallel does to a resident `dashing serve` when DandD's k-batch is NOT taken by dandd_amd/bin/fused/parallel: K client
    processes started together, one listening socket.  All forty are answered, every file is the oracle's."""
    from concurrent.futures import ThreadPoolExecutor
    sock = str(tmp_path / "d.sock")
    env = dict(os.environ)
    env.pop("DANDD_DASHING_SERVER", None)
    env.pop("DANDD_SKETCH_FORMAT", None)
    fasta = os.path.join(GOLD, "fasta", "g1.fasta")
    srv = _start_server(dashing, sock, env)
    try:
        cenv = dict(env, DANDD_DASHING_SERVER=sock, DANDD_SERVER_REQUIRED="1")
        for k in range(1, 41):
            os.makedirs(tmp_path / f"k{k}")

        def one(k):
            return subprocess.run([dashing, "sketch", f"-k{k}", "-S", "12", "--prefix", str(tmp_path / f"k{k}"), fasta], env=cenv, capture_output=True, text=True, timeout=300)
        with ThreadPoolExecutor(max_workers=40) as pool:
            results = list(pool.map(one, range(1, 41)))
        assert all(r.returncode == 0 for r in results), [(r.returncode, r.stderr) for r in results if r.returncode]
        buf = np.fromfile(fasta, dtype=np.uint8)
        want = orc.sketch_sweep(buf, 1, 40, 12, True)
        for k in range(1, 41):
            raw = np.fromfile(tmp_path / f"k{k}" / f"g1.fasta.w.{k}.spacing.12.hll", dtype=np.uint8)
            assert np.array_equal(raw[NATIVE_HEAD:], want[k - 1]), k
        r = subprocess.run([dashing, "shutdown"], env=cenv, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and int(r.stdout) == 40
        assert srv.wait(timeout=60) == 0
    finally:
        if srv.poll() is None:
            srv.kill()
