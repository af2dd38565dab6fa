"""The randomized sweeps of scripts/fuzz_*.py, bounded and seeded, under `-m gpu`: what earlier rounds ran by hand (tens of
thousands of draws) the driver now witnesses a slice of.  Each script draws its configurations from
numpy's default_rng(SEED), checks the HIP path against the oracle (or zlib, or the streaming kernels) bit for bit and exits
non-zero at the first disagreement, printing the draw -- so a failure here names `python scripts/<script> N SEED` to replay."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

# script, draws, seed, what "agree" means
FUZZ = [
    ("fuzz_parity.py", 1500, 505, "K0 + K1 registers == oracle: log2m 4-20, k ranges in 1..64, canonical or not, kseq / FASTQ record shapes"),
    ("fuzz_buckets.py", 1000, 506, "log2m 16-20 scatter / sort / replay path with random schedule knobs == oracle"),
    ("fuzz_inflate.py", 100, 507, "device-inflated BGZF blocks and single gzip members: bytes == the compressed text, registers == plain sketch (strict)"),
    ("fuzz_damage.py", 300, 508, "damaged .gz files: the call raises exactly when zlib's gzread fails, else the registers of gzread's text"),
    ("fuzz_fastq.py", 250, 510, "FASTQ-like texts in .gz / BGZF / two members, clean and broken: whichever of the device's rules or the host's kseq state machine takes a text, registers == the oracle's kseq reading"),
    ("fuzz_k2.py", 50, 509, "Gram all-pairs == streaming kernel, bit-plane progressive scan == streaming kernel"),
    ("fuzz_cli.py", 40, 511, "the command-line boundary through one `dashing serve`: k-batches via dandd_amd/bin/fused/parallel == one `dashing sketch` per k == oracle, unions, multi-path card, three sketch containers"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("script,draws,seed,what", FUZZ, ids=[f[0][:-3] for f in FUZZ])
def test_bounded_seeded_fuzz(torch_cuda, script, draws, seed, what):
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)     # (a failing draw's input is saved there)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", script), str(draws), str(seed)], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    tail = (r.stdout[-1500:] + r.stderr[-1500:])
    assert r.returncode == 0, f"replay: python scripts/{script} {draws} {seed}\n{tail}"
    assert str(draws) in r.stdout.splitlines()[-1], tail          # the script's own summary line: every draw was run
