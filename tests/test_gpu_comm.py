"""Multi-GPU behind the C ABI (include/dandd_hip.h: dd_comm_*, dd_allreduce_max_u8, dd_allgather_u8 -- RCCL called by
libdandd_hip.so itself), SURVEY 8(e).  One GPU per box here, so: a communicator of world size 1 through the Python binding, a plain
C99 rank program (tests/native/comm_ranks.c: no Python, no torch) at world size 1 and -- two processes, two contexts -- at world
size 2 on the one GPU, which RCCL may refuse ("Duplicate GPU"): then the refusal must be a clean error, not a hang or a crash;
and bench.py with its collectives routed through the C ABI."""
import json
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


@pytest.mark.gpu
def test_world1_communicator_through_the_binding(torch_cuda):
    torch = torch_cuda
    from dandd_amd.engine import Engine, EngineError, comm_unique_id
    with Engine(device=0, log2m=14, canonical=True) as eng:
        assert eng.comm_info()[1] == 0
        with pytest.raises(EngineError):
            eng.allreduce_max_u8(0, 0)                              # no communicator yet
        uid = comm_unique_id()
        assert len(uid) == 128 and any(uid)
        eng.set_stream(torch.cuda.current_stream().cuda_stream)
        eng.comm_init(0, 1, uid)
        with pytest.raises(EngineError):
            eng.comm_init(0, 1, uid)                                # one communicator per context
        rng = np.random.default_rng(7)
        host = rng.integers(0, 50, size=(37, 1 << 14), dtype=np.uint8)
        t = torch.from_numpy(host).cuda()
        eng.allreduce_max_u8(t.data_ptr(), t.numel())              # ncclAllReduce(ncclUint8, ncclMax) over one rank: the identity
        out = torch.empty((1,) + tuple(t.shape), dtype=torch.uint8, device="cuda")
        eng.allgather_u8(t.data_ptr(), t.numel(), out.data_ptr())
        eng.synchronize()
        assert np.array_equal(t.cpu().numpy(), host) and np.array_equal(out[0].cpu().numpy(), host)
        assert eng.comm_info() == (0, 1, 1, 1)
        eng.comm_destroy()
        assert eng.comm_info()[1] == 0


def _build_comm_ranks(tmp_path):
    from dandd_amd import engine
    if shutil.which("gcc") is None:
        pytest.skip("gcc needed")
    exe = str(tmp_path / "comm_ranks")
    libdir = os.path.dirname(engine.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
           os.path.join(ROOT, "tests", "native", "comm_ranks.c"), "-o", exe, "-L" + libdir, "-ldandd_hip", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"]
    b = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert b.returncode == 0, b.stderr[-3000:]
    return exe


@pytest.mark.gpu
def test_c_rank_program_world1_and_world2(torch_cuda, tmp_path):
    """A host binding that is not Python gets the multi-GPU path from include/dandd_hip.h alone: tests/native/comm_ranks.c sketches
    its rank's genome, all-gathers the leaf slabs and max-all-reduces the root through the C ABI and checks both against sketches it
    makes itself.  World 1 must pass.  World 2 = two processes on this box's one GPU: passes where RCCL lets two ranks share a
    device; where it refuses, both ranks must come back with the refusal (exit 77) -- promptly, no hang."""
    exe = _build_comm_ranks(tmp_path)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([exe, "0", "1", str(tmp_path / "id1")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "comm_ranks: ok, rank 0 of 1" in r.stdout, (r.returncode, r.stdout, r.stderr[-2000:])
    procs = [subprocess.Popen([exe, str(k), "2", str(tmp_path / "id2")], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for k in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("two ranks on one GPU: hung instead of working or refusing")
        outs.append((p.returncode, o, e))
    codes = sorted(c for c, _, _ in outs)
    if codes == [0, 0]:
        assert all(f"rank {k} of 2" in outs[k][1] for k in range(2)), outs
    else:
        assert set(codes) <= {77}, outs          # refused by RCCL (duplicate GPU): a clean error on both ranks
        import torch
        if torch.cuda.device_count() >= 2:
            raise AssertionError(f"two GPUs are visible and the two-rank job still failed: {outs}")
        pytest.skip("RCCL refuses two ranks on one GPU here: " + outs[0][2].strip()[-200:])


@pytest.mark.gpu
def test_bench_collectives_through_the_c_abi(torch_cuda, tmp_path):
    """bench.py --force-dist --abi-comm: the step's root all-reduce and the leaf all-gather go through dd_allreduce_max_u8 /
    dd_allgather_u8 (torch.distributed only carries the communicator's id); the numbers are the plain run's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DD_BENCH_BACKEND", "DD_BENCH_SHARE_DEVICE")}
    common = ["--steps", "2", "--warmup", "1", "--mbp", "2", "--no-cpu-baseline", "--config", "cfg4share", "--detail", str(tmp_path / "d.json")]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common + extra, env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    plain, abi = run([]), run(["--force-dist", "--abi-comm"])
    c = abi["collectives"]
    assert c["backend"].startswith("rccl via the C ABI") and c["all_reduce_max_u8"] == 3 and c["all_gather"] == 9
    assert abi["schedule"]["last_prefix_equals_root"] is True
    for key in ("delta_root", "argmax_k_root", "delta_genome0", "argmax_k_genome0"):
        assert abi[key] == plain[key], key
