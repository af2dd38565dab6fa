"""`dandd tree | progressive | kij` on the MI355X engine: same sub-commands, flags, defaults and
output files as /root/reference/lib/dandd_cmd.py (flags :141-286, handlers :43-132); only the
sketching backend differs.  Run as  python -m dandd_amd.host.cli <subcommand> ...
"""
import argparse
import os
import pickle
import sys

from . import deltatree
from .deltatree import write_listdict_to_csv


def tree_command(args):
    if not (args.genomedir or args.flist_loc):
        print("ERROR: You must provide either a datadirectory or a fasta file list!")
        sys.exit(1)
    if not args.sketchdir:
        args.sketchdir = os.path.join(args.outdir, "sketchdb")
        os.makedirs(args.sketchdir, exist_ok=True)
    tool = "dashing"
    if args.exact:  # exact distinct k-mer counts on the GPU (the reference's KMC branch, :51-53)
        tool = "kmc"
        args.registers = 20
    ksweep = (int(args.mink), int(args.maxk)) if args.ksweep else None
    os.makedirs(args.outdir, exist_ok=True)
    # a sketch directory with nothing in it: this run will sketch, so the GPU context is brought up beside the
    # digests and directory walks instead of after them (a fully cached run never touches the GPU and starts none)
    try:
        cold = not any(e.is_dir() and any(os.scandir(e.path)) for e in os.scandir(args.sketchdir))
    except OSError:
        cold = True
    if cold:
        deltatree.prewarm_backend({"registers": int(args.registers), "canonicalize": args.canonicalize, "tool": tool})
    try:
        tree = deltatree.create_delta_tree(
            tag=args.tag, genomedir=args.genomedir, sketchdir=args.sketchdir, kstart=args.kstart,
            nchildren=args.nchildren, registers=int(args.registers), flist_loc=args.flist_loc,
            canonicalize=args.canonicalize, tool=tool, debug=args.debug, nthreads=int(args.nthreads),
            safety=args.safety, fast=args.fast, verbose=args.verbose, ksweep=ksweep, lowmem=args.lowmem)
    except deltatree.WorkerDone:  # rank > 0 of a multi-GPU run: its leaf sketches are on disk
        return
    prefix = tree.make_prefix(outdir=args.outdir, tag=args.tag, label=args.label)
    tree.save(fileprefix=prefix, fast=args.fast)


def _load_tree(path):
    """A tree pickle written by this package or by the reference itself (dandd_amd/host/compat.py)."""
    from .compat import load_tree
    return load_tree(path)


def progressive_command(args):
    tree = _load_tree(args.delta_tree)
    if not args.tag:
        args.tag = tree.speciesinfo.tag
    outfile = tree.make_prefix(tag=args.tag, label=f"progu{args.norderings}", outdir=args.outdir)
    exp = tree.experiment
    exp.update(debug=args.debug, safety=args.safety, fast=args.fast, verbose=args.verbose, lowmem=args.lowmem,
               baseset=set(), ksweep=(int(args.mink), int(args.maxk)) if args.ksweep else None)
    tree.speciesinfo.update(tool=exp["tool"])
    if deltatree.dist_ranks()[0] != 0:
        # rank > 0 of a multi-GPU run: its share of the leaf sketches the sweep needs, then done.  The spider
        # over the run's FASTAs is the one rank 0 builds inside progressive_union; both meet at its barrier.
        try:
            deltatree.DeltaSpider(fasta_files=tree.progressive_fastas(args.flist_loc), speciesinfo=tree.speciesinfo,
                                  experiment=exp)
        except deltatree.WorkerDone:
            pass
        return
    results, summary = tree.progressive_wrapper(flist_loc=args.flist_loc, count=args.norderings,
                                                ordering_file=args.ordering_file, step=args.step)
    write_listdict_to_csv(outfile + ".csv", results)
    write_listdict_to_csv(outfile + "summary.csv", summary)
    tree.save(fileprefix=outfile)


def kij_command(args):
    tree = _load_tree(args.delta_tree)
    tree.speciesinfo.update(tool=tree.experiment["tool"])
    if not args.tag:
        args.tag = tree.speciesinfo.tag
    outfile = tree.make_prefix(tag=args.tag, label=args.label, outdir=args.outdir)
    if args.flist_loc:
        with open(args.flist_loc) as f:
            wanted = {line.strip() for line in f}
        sub = [n for n in tree.leaf_nodes() if n.fastas[0] in wanted or os.path.basename(n.fastas[0]) in wanted]
    else:
        sub = []
    if args.ksweep:
        tree.experiment["ksweep"] = (int(args.mink), int(args.maxk))
    if deltatree.dist_ranks()[1] > 1:
        # every rank sketches its share of the leaves for the whole range; rank 0 finishes alone
        rank = deltatree.dist_ranks()[0]
        tree.presketch_range(int(args.mink), int(args.maxk))   # (ends at the barrier, which switches sharding off)
        if rank != 0:
            return
    tree.ksweep(mink=int(args.mink), maxk=int(args.maxk))
    kij_rows, j_rows = tree.pairwise_spiders(sublist=sub, mink=args.mink, maxk=args.maxk, jaccard=args.jaccard)
    write_listdict_to_csv(outfile + ".kij.csv", kij_rows)
    if args.jaccard:
        write_listdict_to_csv(outfile + ".j.csv", j_rows)
    tree.speciesinfo.save_cardkey(tree.experiment["tool"])
    tree.speciesinfo.save_references(fast=False)
    if args.afproject:
        with open(outfile + "_AFtuples.pickle", "wb") as f:
            pickle.dump(tree.prepare_AFproject(kij_rows, j_rows), f)


def serve_command(args):
    """A resident `dandd`: commands arrive over a unix socket (dandd_amd.host.client), run one at a time in THIS process --
    whose backends (deltatree._backends: GPU context, kernel modules, pinned buffers, job-table cache) outlive them -- with the
    client's working directory and DANDD_* / DD_* environment, and their stdout / stderr / exit status go back.  The code that
    runs is main() below, the one-shot CLI's: same files, byte for byte."""
    import contextlib
    import io
    import socket
    import time
    from .client import ENV_PREFIXES, recv_msg, send_msg
    path = args.socket
    try:
        os.unlink(path)
    except FileNotFoundError:
        pass
    srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    old_mask = os.umask(0o177)   # the socket runs commands as this user: nobody else may connect, from the moment it exists
    try:
        srv.bind(path)
    finally:
        os.umask(old_mask)
    srv.listen(8)
    if args.warm:              # bring a backend up before the first command arrives: "<registers>[,nc]"
        for spec in args.warm:
            regs, _, flag = spec.partition(",")
            deltatree.backend_for({"registers": int(regs), "canonicalize": flag != "nc", "tool": "dashing"})
    deltatree.RESIDENT = True
    print(f"dandd serve: listening on {path}", flush=True)
    served, home = 0, os.getcwd()

    def handle(conn):
        """one connection = one request; -> True when the server is asked to leave"""
        nonlocal served
        conn.settimeout(10.0)      # a client that connects and then says nothing (or never reads its reply) holds the single line ten seconds, not for ever
        req = recv_msg(conn)
        if not isinstance(req, dict):
            return False
        if req.get("op") == "ping":
            send_msg(conn, {"rc": 0, "served": served, "pid": os.getpid()})
            return False
        if req.get("op") == "shutdown":
            send_msg(conn, {"rc": 0, "served": served})
            return True
        out, err = io.StringIO(), io.StringIO()
        saved = {k: v for k, v in os.environ.items() if k.startswith(ENV_PREFIXES)}
        t0 = time.perf_counter()
        rc = 1
        try:
            for k in saved:
                del os.environ[k]
            os.environ.update({k: str(v) for k, v in (req.get("env") or {}).items() if k.startswith(ENV_PREFIXES)})
            deltatree.new_command()
            with contextlib.redirect_stdout(out), contextlib.redirect_stderr(err):
                try:
                    os.chdir(req.get("cwd") or home)      # (a directory that is gone: the client gets the message, like any failed command)
                    argv = list(req.get("argv") or [])
                    if argv[:1] == ["serve"]:
                        raise SystemExit("dandd serve: a server does not start servers")
                    rc = main(argv) or 0
                except SystemExit as e:
                    rc = e.code if isinstance(e.code, int) else (0 if e.code is None else 1)
                    if isinstance(e.code, str):
                        print(e.code, file=sys.stderr)
                except BaseException:
                    import traceback
                    traceback.print_exc()
                    rc = 1
        finally:
            for k in [k for k in os.environ if k.startswith(ENV_PREFIXES)]:
                del os.environ[k]
            os.environ.update(saved)
            os.chdir(home)
        served += 1
        send_msg(conn, {"rc": rc, "stdout": out.getvalue(), "stderr": err.getvalue(), "seconds": time.perf_counter() - t0})
        return False

    try:
        while True:
            srv.settimeout(args.idle_exit if args.idle_exit > 0 else None)
            try:
                conn, _ = srv.accept()
            except socket.timeout:
                break
            with conn:
                # a client that hangs up mid-command (Ctrl-C, a timeout) or sends something that is not a request costs its own
                # connection, never the server: the warm GPU context is what this process exists to keep
                try:
                    if handle(conn):
                        break
                except (OSError, ValueError) as e:   # (BrokenPipeError, ConnectionResetError; JSONDecodeError, UnicodeDecodeError)
                    print(f"dandd serve: connection dropped: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
    finally:
        srv.close()
        try:
            os.unlink(path)
        except OSError:
            pass
    return 0


def build_parser():
    common = argparse.ArgumentParser(add_help=False)
    common.add_argument("--version", action="version", version="%(prog)s 1.0.0 (dandd_amd / MI355X)")
    common.add_argument("--verbose", "-v", action="store_true", default=False)
    common.add_argument("--debug", action="store_true", default=False)
    common.add_argument("--lowmem", action="store_true", default=False)
    common.add_argument("--safe", action="store_true", default=False, dest="safety")
    common.add_argument("--fast", action="store_true", default=False)
    sweep = argparse.ArgumentParser(add_help=False)
    sweep.add_argument("--ksweep", dest="ksweep", default=None, action="store_true")
    sweep.add_argument("--mink", dest="mink", default=2, type=int)
    sweep.add_argument("--maxk", dest="maxk", default=32, type=int)

    parser = argparse.ArgumentParser(prog="DandD", parents=[common],
                                     description="delta values for a set of fasta files (MI355X sketching engine)")
    subs = parser.add_subparsers(title="subcommands")
    subs.required = True

    t = subs.add_parser("tree", parents=[common, sweep])
    t.add_argument("-s", "--tag", dest="tag", type=str, default="dandd")
    t.add_argument("-x", "--exact", dest="exact", default=False, action="store_true")
    t.add_argument("-d", "--datadir", dest="genomedir", default=None, type=str)
    t.add_argument("-o", "--out", dest="outdir", default=os.getcwd(), type=str)
    t.add_argument("-c", "--sketchdir", dest="sketchdir", default=None, type=str)
    t.add_argument("-k", "--kstart", dest="kstart", default=12, type=int)
    t.add_argument("-f", "--fastas", dest="flist_loc", type=str, default=None)
    t.add_argument("-l", "--label", dest="label", default="")
    t.add_argument("-n", "--nchildren", dest="nchildren", type=int, default=None)
    t.add_argument("-r", "--registers", dest="registers", default=20)
    t.add_argument("-e", "--nthreads", dest="nthreads", type=int, default=0)
    t.add_argument("-C", "--no-canon", action="store_false", default=True, dest="canonicalize")
    t.set_defaults(func=tree_command)

    p = subs.add_parser("progressive", parents=[common, sweep])
    p.add_argument("-d", "--dtree", dest="delta_tree", required=True)
    p.add_argument("-s", "--tag", dest="tag", type=str)
    p.add_argument("-r", "--orderings", dest="ordering_file", type=str, default=None)
    p.add_argument("-f", "--fastas", dest="flist_loc", default=None, type=str)
    p.add_argument("-n", "--norderings", dest="norderings", default=0, type=int)
    p.add_argument("-o", "--outdir", dest="outdir", default=os.getcwd(), type=str)
    p.add_argument("-l", "--label", dest="label", default="")
    p.add_argument("--step", dest="step", default=1, type=int)
    p.set_defaults(func=progressive_command)

    k = subs.add_parser("kij", parents=[common, sweep])
    k.add_argument("-d", "--dtree", dest="delta_tree", required=True)
    k.add_argument("-s", "--tag", dest="tag", type=str)
    k.add_argument("-f", "--fastas", dest="flist_loc", default=None, type=str)
    k.add_argument("-o", "--outdir", dest="outdir", default=os.getcwd(), type=str)
    k.add_argument("-l", "--label", dest="label", default="")
    k.add_argument("--afproject", dest="afproject", default=False, action="store_true")
    k.add_argument("--jaccard", dest="jaccard", default=False, action="store_true")
    k.set_defaults(func=kij_command)

    # (not in the reference: its every command is a fresh process that shells out to fresh `dashing` processes)
    sv = subs.add_parser("serve", description="keep the GPU context alive and run the commands dandd_amd.host.client forwards")
    sv.add_argument("--socket", required=True, help="path of the unix socket to listen on (clients: DANDD_SERVER=<path>)")
    sv.add_argument("--idle-exit", type=float, default=0.0, help="leave after this many seconds without a command (0: never)")
    sv.add_argument("--warm", action="append", default=[], metavar="REGISTERS[,nc]", help="create this backend before listening, e.g. --warm 20")
    sv.set_defaults(func=serve_command)
    return parser


def main(argv=None):
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        # one process per GPU (torch.distributed.run): every sub-command shards the leaf sketches it needs
        # over the ranks and hands them over through the shared sketch directory; the process group is only
        # a barrier
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo", rank=rank, world_size=world)
        deltatree.set_dist_active(True)
    elif "torch" not in sys.modules:  # stand-alone process: no torch anywhere on this path (engine.load_library)
        os.environ.setdefault("DANDD_NO_TORCH", "1")
    try:
        args = build_parser().parse_args(sys.argv[1:] if argv is None else argv)
        args.func(args)
        if dist is not None:
            dist.barrier()
    except BaseException as e:
        if dist is not None and not (isinstance(e, SystemExit) and not e.code):
            # a rank that fails must not leave its peers waiting in a barrier it will never reach: leave at once with
            # a non-zero status and let the launcher (torch.distributed.run) end the others
            import traceback
            traceback.print_exc()
            sys.stderr.flush()
            sys.stdout.flush()
            os._exit(1)
        raise
    finally:
        deltatree.set_dist_active(False)
        if dist is not None:
            dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
