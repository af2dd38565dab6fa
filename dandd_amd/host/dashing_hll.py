"""Export / import of sketches in Dashing's own `.hll` container, so that a
sketch directory written by this engine can be handed to a real `dashing card|union|dist` and back.

STATUS: UNVERIFIED.  No Dashing binary or source exists in this image or in /root/reference, so the
layout below is the published one as recalled (dnbaker/sketch `hll.h`, `hllbase_t::write/read`, the
version Dashing 0.4-1.0 vendors), not one that was checked against the program:

    gzip stream of
        uint32  is_calculated     (0: the cached estimate below is not valid)
        uint32  clamp             (0)
        uint32  estimator         (0 ORIGINAL, 1 ERTL_IMPROVED, 2 ERTL_MLE       -- Dashing's default: 2)
        uint32  joint estimator   (0 ORIGINAL, 1 ERTL_IMPROVED, 2 ERTL_MLE, 3 ERTL_JOINT_MLE -- default: 3)
        uint32  nthreads          (1)
        uint32  np                (log2 of the register count)
        double  cached estimate
        uint8   registers[2^np]   (same values as this engine's: rho of the 64 - np low hash bits)

The register bytes themselves are what this repo verifies bit for bit against its oracle; only the
32-byte header is recalled.  k and the canonical flag are not part of Dashing's container (they live in
the file NAME, `<fasta>.w.<k>.spacing.<R>.hll`), so reading one back needs them from the caller.

    python -m dandd_amd.host.dashing_hll export <sketch.hll> <out.hll>      (native -> Dashing)
    python -m dandd_amd.host.dashing_hll import <in.hll> <sketch.hll> K [--no-canon]
"""
import sys

from .backend import read_sketch_file, write_sketch_file


def write_dashing_hll(path, regs, log2m, estimate=None, compressed=True):
    """(the cached estimate slot is left invalid: Dashing recomputes it)"""
    write_sketch_file(path, regs, log2m, 0, True, fmt="dashing" if compressed else "dashing-plain")


def read_dashing_hll(path):
    """-> (registers uint8[2^np], np, None)"""
    regs, np_, _k, _canon = read_sketch_file(path)
    return regs, np_, None


def main(argv):
    if len(argv) >= 3 and argv[0] == "export":
        regs, log2m, _k, _canon = read_sketch_file(argv[1])
        write_dashing_hll(argv[2], regs, log2m)
        return 0
    if len(argv) >= 4 and argv[0] == "import":
        regs, log2m, _ = read_dashing_hll(argv[1])
        write_sketch_file(argv[2], regs, log2m, int(argv[3]), "--no-canon" not in argv, fmt="native")
        return 0
    print(__doc__)
    return 2


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
