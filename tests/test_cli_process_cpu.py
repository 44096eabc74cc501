"""The CLI's process model (cli.cpp main): one process by default; with SBX_DETACH=1 the work runs in a child and the command
returns with the child's status once the output is complete.  Without a GPU only the failure paths can run -- they must look
the same in both modes."""
import os
import subprocess

import pytest

from sambamba_amd import cli_path


def run(args, **env):
    e = dict(os.environ, **env)
    return subprocess.run([cli_path()] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=e, timeout=120)


@pytest.mark.parametrize("args", [["base", "/nonexistent/x.bam"], ["region", "/nonexistent/x.bam"], ["window", "/nonexistent/x.bam"],
                                  ["base", "--no-such-option", "x.bam"], []])
def test_status_and_streams_equal_in_both_process_modes(args):
    a = run(args, SBX_DETACH="1")
    b = run(args)
    assert a.returncode == b.returncode
    assert a.stdout == b.stdout
    assert a.stderr == b.stderr
    if args:
        assert a.returncode == 1 and a.stderr.startswith(b"sambamba-depth: ") or b"must be provided" in a.stderr


def test_killed_child_is_reported(tmp_path):
    """A child that dies without a report (here: SIGKILL through a tiny address-space limit is not portable, so a signal
    sent to the process group) must not make the command succeed."""
    import signal
    import time
    p = subprocess.Popen([cli_path(), "base", "/dev/stdin"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         start_new_session=True, env=dict(os.environ, SBX_DETACH="1"))
    time.sleep(0.3)
    try:
        os.killpg(p.pid, signal.SIGKILL)
    except ProcessLookupError:
        pass
    p.communicate()
    assert p.returncode != 0


def test_a_child_killed_by_a_signal_kills_the_command_by_that_signal(tmp_path):
    """Detached: the worker child dies of SIGTERM (sent to it alone) before it could report -- the command must not turn that
    into an exit status, it dies by the same signal (a shell sees 128 + 15, subprocess sees -15)."""
    import signal
    import time
    fifo = str(tmp_path / "never_written.bam")
    os.mkfifo(fifo)                 # opening it blocks the worker for as long as nobody writes
    p = subprocess.Popen([cli_path(), "base", fifo], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         env=dict(os.environ, SBX_DETACH="1"))
    kids, t0 = [], time.time()
    while not kids and time.time() - t0 < 0.5 and p.poll() is None:      # (the worker may live for milliseconds only where there is no device)
        kids = subprocess.run(["pgrep", "-P", str(p.pid)], stdout=subprocess.PIPE, universal_newlines=True).stdout.split()
    if kids:
        try:
            os.kill(int(kids[0]), signal.SIGSTOP)        # hold it wherever it is, then deliver the fatal signal
        except ProcessLookupError:
            kids = []
    if not kids:
        p.kill()
        p.communicate()
        pytest.skip("the worker had already ended")
    os.kill(int(kids[0]), signal.SIGTERM)
    os.kill(int(kids[0]), signal.SIGCONT)
    p.communicate()
    assert p.returncode == -signal.SIGTERM
