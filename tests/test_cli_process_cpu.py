"""The CLI's process model (cli.cpp main): the work runs in a child, the command returns with the child's status once the
output is complete.  Without a GPU only the failure paths can run -- they must look the same in both modes."""
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
    a = run(args)
    b = run(args, SBX_NO_DETACH="1")
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
                         start_new_session=True)
    time.sleep(0.3)
    try:
        os.killpg(p.pid, signal.SIGKILL)
    except ProcessLookupError:
        pass
    p.communicate()
    assert p.returncode != 0
