"""End-to-end timing of the drop-in CLI (tree -> progressive -> kij) on synthetic genomes written as
FASTA files, i.e. everything a DandD user pays for: Python start-up, file reads, H2D copies, K0..K3,
the host-side tree / spider logic, pickles and CSVs.  Development aid, not the contract bench.

  python scripts/e2e_cli.py NGENOMES MBP [--registers P] [--mink A --maxk B] [--norderings N] [--dir D]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("ngenomes", type=int)
    ap.add_argument("mbp", type=float)
    ap.add_argument("--registers", type=int, default=14)
    ap.add_argument("--mink", type=int, default=4)
    ap.add_argument("--maxk", type=int, default=40)
    ap.add_argument("--norderings", type=int, default=10)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--profile", default=None, help="sub-command to run under cProfile (tree|progressive|kij)")
    ap.add_argument("--hillclimb", action="store_true", help="no --ksweep: DandD's default argmax-k search from -k 12")
    ap.add_argument("--server", action="store_true", help="round 5: run every command TWICE through a resident `dandd serve` (dandd_amd.host.client): the first "
                                                          "pays the context's bring-up inside the server, the second is the warm figure; sketch directory wiped in between")
    ap.add_argument("--gz", type=int, default=None, help="write the genomes as .fasta.gz (one gzip member, this zlib level): what genome directories really hold")
    args = ap.parse_args()

    work = args.dir or tempfile.mkdtemp(prefix="dandd_e2e_")
    gdir, out = os.path.join(work, "genomes"), os.path.join(work, "out")
    os.makedirs(gdir, exist_ok=True)
    os.makedirs(out, exist_ok=True)
    nb = int(args.mbp * 1e6)

    t0 = time.time()
    import torch
    from dandd_amd.engine import Engine, synth_size
    eng = Engine(0, args.registers, True)
    n = synth_size(nb, 5)
    buf = torch.empty(n + 16, dtype=torch.uint8, device="cuda")
    for g in range(args.ngenomes):
        eng.synth_fasta_device(0xD4ADD, g, nb, 5, buf.data_ptr())
        eng.synchronize()
        if args.gz is None:
            buf[:n].cpu().numpy().tofile(os.path.join(gdir, f"g{g:03d}.fasta"))
        else:
            import zlib
            co = zlib.compressobj(args.gz, zlib.DEFLATED, 31)
            with open(os.path.join(gdir, f"g{g:03d}.fasta.gz"), "wb") as f:
                f.write(co.compress(buf[:n].cpu().numpy().tobytes()) + co.flush())
    del eng, buf
    t_gen = time.time() - t0

    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cli = [sys.executable, "-m", "dandd_amd.host.cli"]
    sweep = [] if args.hillclimb else ["--ksweep", "--mink", str(args.mink), "--maxk", str(args.maxk)]
    timings = {}

    def run(name, cmd):
        if args.profile == name:
            prof = os.path.join(work, name + ".prof")
            cmd = [cmd[0], "-m", "cProfile", "-o", prof] + cmd[1:]
        t = time.time()
        r = subprocess.run(cmd, env=env, cwd=work, capture_output=True, text=True)
        timings[name] = round(time.time() - t, 3)
        if r.returncode:
            print(r.stdout[-2000:], r.stderr[-4000:], file=sys.stderr)
            raise SystemExit(f"{name} failed with {r.returncode}")
        if args.profile == name:
            import pstats
            pstats.Stats(os.path.join(work, name + ".prof")).sort_stats("cumulative").print_stats(45)

    server = None
    if args.server:
        sock = os.path.join(work, "dandd.sock")
        server = subprocess.Popen(cli + ["serve", "--socket", sock, "--idle-exit", "300"], env=env, cwd=work, stdout=subprocess.PIPE,
                                  stderr=open(os.path.join(work, "server.err"), "w"), text=True)
        assert "listening" in server.stdout.readline()
        env = dict(env, DANDD_SERVER=sock, DANDD_SERVER_REQUIRED="1")
        cli = [sys.executable, "-m", "dandd_amd.host.client"]
    try:
        if server is not None:
            # first pass = cold server (context bring-up inside it); outputs and sketches wiped; second pass below = warm
            measure(args, run, cli, gdir, out, sweep, timings)
            cold = dict(timings)
            shutil.rmtree(out)
            os.makedirs(out)
            timings.clear()
            timings["cold_server_first_pass"] = cold
        measure(args, run, cli, gdir, out, sweep, timings)
    finally:
        if server is not None:
            from dandd_amd.host.client import request
            request(env["DANDD_SERVER"], {"op": "shutdown"}, timeout=10)
            try:
                server.wait(timeout=30)
            except subprocess.TimeoutExpired:
                server.kill()
    gbp = args.ngenomes * nb / 1e9
    print(json.dumps({
        "mode": "resident server (dandd serve + dandd_amd.host.client), warm" if args.server else "one-shot processes",
        "workload": f"{args.ngenomes} x {args.mbp:g} Mbp synthetic FASTA files, log2m {args.registers}, "
                    f"k {args.mink}-{args.maxk}, {args.norderings} orderings",
        "fasta_generation_s": round(t_gen, 3), "seconds": timings,
        "tree_gbp_per_s_end_to_end": round(gbp / timings["tree"], 3),
        "outputs": sorted(os.listdir(out))[:12]}))
    if not args.keep and not args.dir:
        shutil.rmtree(work, ignore_errors=True)


def measure(args, run, cli, gdir, out, sweep, timings):
    run("tree", cli + ["tree", "-d", gdir, "-o", out, "-s", "e2e", "-r", str(args.registers)] + sweep)
    pick = [f for f in os.listdir(out) if f.endswith(".pickle") and "dtree" in f]
    if not pick:
        pick = [f for f in os.listdir(out) if f.endswith(".pickle")]
    dtree = os.path.join(out, sorted(pick)[0])
    run("tree_again_cached", cli + ["tree", "-d", gdir, "-o", out, "-s", "e2e", "-r", str(args.registers)] + sweep)
    run("progressive", cli + ["progressive", "-d", dtree, "-o", out, "-n", str(args.norderings)] + sweep)
    run("kij", cli + ["kij", "-d", dtree, "-o", out] + (["--jaccard"] + sweep if sweep else []))



if __name__ == "__main__":
    main()
