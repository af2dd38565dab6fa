"""Worker of tests/test_distributed.py::test_two_rank_cli_tree: one rank of `torch.distributed.run ...
cli tree`, with the GPU backend replaced by the CPU checker (tests/hostcheck.OracleBackend plus the
batch entry point the sharded leaf step needs) so the multi-rank host logic runs without a GPU."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import hostcheck  # noqa: E402
from dandd_amd.host import cli, deltatree  # noqa: E402

touched = []


class Backend(hostcheck.OracleBackend):
    def leaf_many(self, fastas, kmin, kmax, path_of):
        for i, f in enumerate(fastas):
            touched.append(os.path.basename(f))
            ks = list(range(kmin, kmax + 1))
            self.leaf(f, ks, [path_of(i, k) for k in ks])


def main():
    log_path = sys.argv[1]
    deltatree.set_backend_factory(Backend)
    cli.main(sys.argv[2:])
    with open(f"{log_path}.{os.environ.get('RANK', '0')}", "w") as f:
        json.dump({"touched": touched}, f)


if __name__ == "__main__":
    main()
