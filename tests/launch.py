"""Subprocess launcher for the multi-rank end-to-end tests: a free rendezvous port, a bounded wait, and -- if the launch does not
finish in time -- SIGABRT to the whole process group so that every rank's faulthandler prints where it stands before the test fails."""
import os
import signal
import socket
import subprocess


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(cmd, cwd, env=None, timeout=300):
    """-> (returncode or None on timeout, stdout, stderr)."""
    env = dict(os.environ if env is None else env, PYTHONFAULTHANDLER="1")
    p = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
        return p.returncode, out, err
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGABRT)          # faulthandler: Python tracebacks of all threads, in every rank
        except ProcessLookupError:
            pass
        try:
            out, err = p.communicate(timeout=30)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, signal.SIGKILL)
            out, err = p.communicate()
        return None, out, "TIMED OUT after %d s\n" % timeout + err
