"""Stream ordering, checked without a device: the CPU build of the engine with the stand-in runtime's streams as real
queues (tests/helpers/hipcpu/hip/hip_runtime.h, HIPCPU_STREAMS):

  lazy    queued work runs only when somebody waits for it -- a missing fence, event wait or synchronisation reads
          memory that nobody has written yet (shown by mutation: without the copy thread's event wait
          mi355_factor_numeric_host returns the previous factor; without the synchronisation in front of a blocking
          collective the partitioned factor is wrong);
  greedy  every wait first runs all OTHER streams as far as they can go -- work a reader should have been protected
          from has already run when it reads.

The RCCL path of the N-rank driver runs here too: the CPU build exports a stand-in for librccl (collectives between
threads, executed as operations of the emulated streams), so the event fences between the compute and the communication
stream are exercised -- without either of its two hipStreamWaitEvent calls the factor is wrong under "lazy" (mutations;
invisible with synchronous streams).

The default suite runs with synchronous streams (fast); here a subset that exercises every stream the engine owns
(compute, side, copy, communication) is rerun in both modes, each in a process of its own (the mode is
read once per process).  The whole emulator suite under both modes: profiles/r02_asan_cpuemu.txt."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SUBSET = ["tests/test_abi_on_cpu.py", "tests/test_multi_abi.py", "-k",
          "(numeric_host and (lap2d_16_metis or tcov)) or test_solve2_with_permutation_all_systems or test_blas_timer "
          "or (threads_as_ranks and (lap3d-4-2 or lap2d-16-4)) or (wide_top and 2-True-1-1-False) "
          "or (rccl_stand_in and (lap3d-6-3 or lap3d-4-2-True-True-64-1 or ob10-1)) or 1-rccl or 3-rccl"]


@pytest.mark.parametrize("mode", ["lazy", "greedy"])
def test_subset_under_queued_streams(mode):
    if not os.path.exists(os.path.join(ROOT, "tests", "helpers", "libmi355_cpuemu.so")):
        pytest.skip("no CPU build of the engine here")
    env = dict(os.environ, HIPCPU_STREAMS=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider"] + SUBSET,
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    tail = "\n".join(r.stdout.splitlines()[-15:])
    assert r.returncode == 0, tail
    assert " passed" in tail and "no tests ran" not in tail, tail
