"""Reads of memory nobody wrote.  A kernel that reads a word of a work area or of an output buffer before anything stored it computes,
in an ordinary run, with whatever the allocation held before -- usually zeros or the previous multiply's values of the same kind, so
the result is right by accident and goes wrong once in a thousand runs, in another order of calls (this is how it showed: two failures
of the long randomised walk when four pytest workers shared the GPU, none in one process).  Here every work area of the engine
(lab build: DBCSR_AMD_MM_POISON) and every torch.empty of the host side is filled with a byte pattern first -- 0xFF: -1 as an index,
NaN as a value; 0xCD: a huge negative index, -6.3e66 as a value -- and the randomised sweep's cases must still match the oracle."""
import os

import pytest
import torch

from tests import test_gpu_random_sweep as sweep

pytestmark = pytest.mark.gpu

N_PLAIN = int(os.environ.get("DBCSR_AMD_POISON_PLAIN", "90"))
N_FORCED = int(os.environ.get("DBCSR_AMD_POISON_FORCED", "48"))


@pytest.fixture
def poisoned(monkeypatch, request):
    byte = request.param
    monkeypatch.setenv("DBCSR_AMD_LAB", "1")
    monkeypatch.setenv("DBCSR_AMD_MM_POISON", str(byte))
    real_empty, real_empty_like = torch.empty, torch.empty_like

    def fill(t):
        if t.is_cuda and t.numel() and t.is_contiguous():
            t.view(torch.uint8).fill_(byte)
        return t

    monkeypatch.setattr(torch, "empty", lambda *a, **k: fill(real_empty(*a, **k)))
    monkeypatch.setattr(torch, "empty_like", lambda *a, **k: fill(real_empty_like(*a, **k)))
    yield byte
    monkeypatch.delenv("DBCSR_AMD_MM_POISON")
    from dbcsr_amd.multiply import MultiplyEngine
    MultiplyEngine(lab=True)   # (an engine created without the variable switches the filling off again for this process)


@pytest.mark.parametrize("poisoned", [255, 205], indirect=True, ids=["ff", "cd"])
@pytest.mark.parametrize("seed", range(N_PLAIN))
def test_random_multiply_on_poisoned_memory(seed, poisoned):
    sweep.run_case(sweep.make_case(1000 + 7 * seed + (poisoned & 1)))


@pytest.mark.parametrize("poisoned", [255], indirect=True, ids=["ff"])
@pytest.mark.parametrize("seed", range(N_FORCED))
def test_forced_paths_on_poisoned_memory(seed, poisoned, monkeypatch):
    for k in ("DBCSR_AMD_MM_CLASSES", "DBCSR_AMD_MM_SYMBOLIC", "DBCSR_AMD_MM_WG_WAVES", "DBCSR_AMD_MM_HOT"):
        monkeypatch.delenv(k, raising=False)
    for k, v in sweep.FORCED[seed % len(sweep.FORCED)].items():
        monkeypatch.setenv(k, v)
    sweep.run_case(sweep.make_case(5000 + 5 * seed))
