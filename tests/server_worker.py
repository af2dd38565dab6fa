"""Starts `dandd serve` with the CPU checker backend behind it (tests/hostcheck.py: OracleBackend), or runs one command in
this process with the same backend: the two sides of tests/test_server.py's comparison.  usage: server_worker.py serve SOCKET | run ARGV..."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import hostcheck  # noqa: E402
from dandd_amd.host import cli, deltatree  # noqa: E402

if os.environ.get("SERVER_WORKER_BACKEND", "oracle") == "oracle":
    deltatree.set_backend_factory(lambda registers, canon: hostcheck.ScheduleBackend(registers, canon))
if sys.argv[1] == "serve":
    sys.exit(cli.main(["serve", "--socket", sys.argv[2], "--idle-exit", "120"]))
sys.exit(cli.main(sys.argv[2:]))
