"""`dandd serve` + dandd_amd.host.client: a resident process runs the forwarded commands with its backends kept alive; what it
writes must be what the one-shot CLI writes, byte for byte (the reference's every command is a fresh process:
/root/reference/lib/dandd_cmd.py:43-132).  CPU: the oracle-backed checker backend on both sides.  GPU: the product backend."""
import filecmp
import os
import shutil
import subprocess
import sys
import time

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = os.path.join(HERE, "golden")
WORKER = os.path.join(HERE, "server_worker.py")


def _commands(out, data, regs):
    tree = ["tree", "-d", data, "-o", out, "-s", "t", "-k", "10", "-r", str(regs), "-c", os.path.join(out, "sketchdb")]
    dtree = os.path.join(out, "t_5_dashing_dtree.pickle")
    return [tree,
            ["progressive", "-d", dtree, "-o", out, "-n", "2", "-r", os.path.join(GOLD, "fasta_orderings.txt")] if os.path.exists(os.path.join(GOLD, "fasta_orderings.txt"))
            else ["progressive", "-d", dtree, "-o", out, "-f", os.path.join(out, "subset.txt"), "-n", "1", "--ksweep", "--mink", "9", "--maxk", "12"],
            ["kij", "-d", dtree, "-o", out, "--jaccard", "--mink", "9", "--maxk", "12"]]


def _prepare(base):
    data = os.path.join(base, "data")
    shutil.copytree(os.path.join(GOLD, "fasta"), data)
    return data


def _files(d):
    out = {}
    for dirpath, _, files in os.walk(d):
        for f in files:
            out[os.path.relpath(os.path.join(dirpath, f), d)] = os.path.join(dirpath, f)
    return out


def _run_both(tmp_path, backend_env, regs, timeout=600):
    """the same three commands one-shot (fresh process each) under tmp/one and through a server under tmp/srv, the data at the
    SAME absolute path for both (moved in and out: pickles and CSVs hold absolute paths)"""
    base = str(tmp_path)
    work = os.path.join(base, "work")
    # (the sketchdb listing and the tree pickle iterate Python SETS of strings, as the reference's do -- /root/reference/lib/
    # huffman_dandd.py:18-21 --: their order follows the process's string-hash seed, so two one-shot runs already differ unless
    # the seed is pinned; pinned, the server's files must be the one-shot CLI's byte for byte)
    env = dict(os.environ, PYTHONHASHSEED="0", **backend_env)
    env.pop("DANDD_SERVER", None)
    results = {}
    for mode in ("one", "srv"):
        os.makedirs(work)
        data = _prepare(work)
        out = os.path.join(work, "out")
        os.makedirs(out)
        names = sorted(os.listdir(data))
        with open(os.path.join(out, "subset.txt"), "w") as f:
            f.write("\n".join(os.path.join(data, n) for n in names[:4]) + "\n")
        times = []
        if mode == "srv":
            sock = os.path.join(base, "dandd.sock")
            srv = subprocess.Popen([sys.executable, WORKER, "serve", sock], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            try:
                line = srv.stdout.readline()
                assert "listening" in line, line + srv.stderr.read()
                cenv = dict(env, DANDD_SERVER=sock, DANDD_SERVER_REQUIRED="1")
                for argv in _commands(out, data, regs):
                    t0 = time.perf_counter()
                    r = subprocess.run([sys.executable, "-m", "dandd_amd.host.client"] + argv, env=cenv, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
                    times.append(time.perf_counter() - t0)
                    assert r.returncode == 0, (argv, r.stdout[-2000:], r.stderr[-3000:])
                # the sketch directory vanishes between two commands (what the server remembers about the file system is true for
                # one command only): the same tree again, from nothing, same rows
                deltas = os.path.join(out, "t_5_dashing_deltas.csv")
                before = open(deltas, "rb").read()
                shutil.rmtree(os.path.join(out, "sketchdb"))
                os.remove(deltas)
                r = subprocess.run([sys.executable, "-m", "dandd_amd.host.client"] + _commands(out, data, regs)[0], env=cenv, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
                assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
                assert open(deltas, "rb").read() == before
                for argv in _commands(out, data, regs)[1:]:      # (and the files the comparison below looks at, rewritten over the new sketches)
                    r = subprocess.run([sys.executable, "-m", "dandd_amd.host.client"] + argv, env=cenv, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
                    assert r.returncode == 0, (argv, r.stdout[-2000:], r.stderr[-3000:])
                # a command that fails in the server comes back as a status and a message, and the server lives on
                r = subprocess.run([sys.executable, "-m", "dandd_amd.host.client", "tree", "-o", out], env=cenv, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
                assert r.returncode == 1 and "ERROR: You must provide" in r.stdout
                # clients that send something that is not a request, or hang up in the middle of one, cost their own connection only
                import socket as socketlib
                for junk in (b"\x05\x00\x00\x00notjs", b"\x40\x00\x00\x00{\"op\": \"run\"", b"\x02\x00\x00\x00\xff\xfe", b"\x04\x00\x00\x00[1]\n"):
                    c = socketlib.socket(socketlib.AF_UNIX, socketlib.SOCK_STREAM)
                    c.connect(sock)
                    c.sendall(junk)
                    c.close()
                assert oct(os.stat(sock).st_mode & 0o777) == "0o600"
                from dandd_amd.host.client import request
                gone = request(sock, {"argv": ["tree", "-o", out], "cwd": os.path.join(base, "no", "such", "dir"), "env": {}})
                assert gone["rc"] == 1 and "FileNotFoundError" in gone["stderr"]      # (a working directory that is gone: a message, not a dropped line)
                assert request(sock, {"op": "ping"})["served"] == 8
                assert request(sock, {"op": "shutdown"})["rc"] == 0
                srv.wait(timeout=60)
            finally:
                if srv.poll() is None:
                    srv.kill()
        else:
            for argv in _commands(out, data, regs):
                t0 = time.perf_counter()
                r = subprocess.run([sys.executable, WORKER, "run"] + argv, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
                times.append(time.perf_counter() - t0)
                assert r.returncode == 0, (argv, r.stdout[-2000:], r.stderr[-3000:])
        kept = os.path.join(base, mode)
        os.rename(work, kept)
        results[mode] = (kept, times)
    return results


def _assert_same_outputs(results):
    one, srv = _files(os.path.join(results["one"][0], "out")), _files(os.path.join(results["srv"][0], "out"))
    assert set(one) == set(srv) and len(one) > 10, (sorted(set(one) ^ set(srv)), len(one))
    differing = [n for n in sorted(one) if not filecmp.cmp(one[n], srv[n], shallow=False)]
    # (sketch files of this package's own container hold no time stamps; gzip'd Dashing containers would)
    assert not differing, differing
    assert any(n.endswith(".kij.csv") for n in one) and any(n.endswith("_deltas.csv") for n in one) and any(n.endswith("summary.csv") for n in one)


def test_server_outputs_equal_one_shot_cli_with_checker_backend(tmp_path):
    _assert_same_outputs(_run_both(tmp_path, {"SERVER_WORKER_BACKEND": "oracle"}, 12))


def test_client_without_a_server_runs_the_command_itself(tmp_path):
    env = dict(os.environ, DANDD_SERVER=str(tmp_path / "nobody.sock"))
    r = subprocess.run([sys.executable, "-m", "dandd_amd.host.client", "tree", "-o", str(tmp_path)], env=env, capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 1 and "ERROR: You must provide" in r.stdout          # the one-shot CLI's own answer
    r = subprocess.run([sys.executable, "-m", "dandd_amd.host.client", "tree", "-o", str(tmp_path)], env=dict(env, DANDD_SERVER_REQUIRED="1"),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 111 and "no server" in r.stderr


@pytest.mark.gpu
def test_server_outputs_equal_one_shot_cli_on_the_gpu(tmp_path, torch_cuda):
    res = _run_both(tmp_path, {"SERVER_WORKER_BACKEND": "hip"}, 14)
    _assert_same_outputs(res)
