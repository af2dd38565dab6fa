"""`dandd` against a resident server: the same sub-commands and flags as the one-shot CLI (dandd_amd.host.cli, i.e.
/root/reference/lib/dandd_cmd.py:141-286), run by a process that keeps its GPU context, kernel modules and pinned buffers
alive between commands.

    python -m dandd_amd.host.cli serve --socket /run/user/1000/dandd.sock &      # once
    DANDD_SERVER=/run/user/1000/dandd.sock python -m dandd_amd.host.client tree -d genomes/ -o out/ ...

A one-shot `dandd tree` on 10 x 50 Mbp spends 0.68 s around 14-28 ms of GPU work: interpreter and imports, hipInit, the first
launch of every kernel module, the driver's tear-down at exit.  None of that is paid by a command that
is forwarded.  Outputs are written by the same code in the server process, with the client's working directory and its
DANDD_* / DD_* environment: byte-identical files.  Without a reachable server the command runs in this process as always.
This module imports nothing heavy (no numpy, no ctypes): a forwarded command costs the interpreter's start and a socket."""
import json
import os
import socket
import struct
import sys

ENV_PREFIXES = ("DANDD_", "DD_")


def send_msg(sock, obj):
    data = json.dumps(obj).encode()
    sock.sendall(struct.pack("<I", len(data)) + data)


def recv_msg(sock):
    head = b""
    while len(head) < 4:
        part = sock.recv(4 - len(head))
        if not part:
            return None
        head += part
    (n,) = struct.unpack("<I", head)
    data = bytearray()
    while len(data) < n:
        part = sock.recv(min(1 << 20, n - len(data)))
        if not part:
            return None
        data += part
    return json.loads(bytes(data))


def request(path, obj, timeout=None):
    """One request to the server at `path`; None when nobody listens there."""
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    try:
        s.settimeout(timeout)
        s.connect(path)
        send_msg(s, obj)
        return recv_msg(s)
    except (FileNotFoundError, ConnectionRefusedError, socket.timeout):
        return None
    finally:
        s.close()


def forward(argv, path):
    """Run `argv` in the server at `path`; its exit status, or None when there is no server (the caller runs it itself)."""
    env = {k: v for k, v in os.environ.items() if k.startswith(ENV_PREFIXES) and k != "DANDD_SERVER"}
    reply = request(path, {"op": "run", "argv": list(argv), "cwd": os.getcwd(), "env": env})
    if reply is None:
        return None
    sys.stdout.write(reply.get("stdout", ""))
    sys.stderr.write(reply.get("stderr", ""))
    return int(reply.get("rc", 1))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    path = os.environ.get("DANDD_SERVER")
    if path and argv and argv[0] != "serve":
        rc = forward(argv, path)
        if rc is not None:
            return rc
        if os.environ.get("DANDD_SERVER_REQUIRED") == "1":
            sys.stderr.write(f"dandd: no server at {path}\n")
            return 111
    from .cli import main as run_here
    return run_here(argv)


if __name__ == "__main__":
    sys.exit(main())
