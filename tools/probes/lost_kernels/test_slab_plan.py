"""Host logic of the 17 - 64-row slab projections (vdd_skinny_slab.hip, laboratory code since round 6): the cut of a projection over the chip, read through the
C ABI's workspace query (no GPU needed: without a device the planner assumes the MI355X's 256 CUs)."""
import ctypes as C

import pytest

import lost_ops

pytestmark = pytest.mark.probe

TICKETS = 16384 * 4        # fixed ticket / counter region at the head of the workspace


def ws_bytes(M, N, K, swiglu=0):
    lib = lost_ops._lib_ready()
    lib.vdd_skinny_slab_workspace_bytes.argtypes, lib.vdd_skinny_slab_workspace_bytes.restype = [C.c_int] * 4, C.c_int64
    return lib.vdd_skinny_slab_workspace_bytes(M, N, K, swiglu)


def slabs(M, N, K, swiglu=0):
    """K parts the planner chose, recovered from the workspace size: tickets + tiles x KS x M-tiles x (gate, up) x 1 KiB."""
    per = ((N + 15) // 16) * ((M + 15) // 16) * (2 if swiglu else 1) * 1024
    body = ws_bytes(M, N, K, swiglu) - TICKETS
    assert body % per == 0
    return body // per


@pytest.mark.parametrize("M,N,K,swiglu,want", [
    (64, 12288, 4096, 0, 4),      # qkv at 64 rows: a 1,024-deep slab of 64 rows is 128 KiB of LDS; 12 tiles per team = 3 per wave
    (18, 12288, 4096, 0, 4),      # LDS would hold all of K at 18 rows, but 3 tiles per team leave a wave idle: cut K four ways
    (64, 4096, 4096, 0, 4),       # attention output: 4 tiles per team, one per wave
    (64, 4096, 11008, 0, 16),     # MLP down at 64 rows: a slab must stay under ~1,150 elements
    (18, 4096, 11008, 0, 4),
    (64, 11008, 4096, 1, 4),      # gate/up (N = features)
    (34, 15360, 5120, 0, 4),      # LLaVA-1.5-13B qkv at the per-rank batch of an 8-GPU split of config #3
])
def test_the_cut_of_the_llava_projections(M, N, K, swiglu, want):
    assert slabs(M, N, K, swiglu) == want


def test_shapes_that_are_not_served():
    assert ws_bytes(65, 4096, 4096) == -1            # more than four M tiles
    assert ws_bytes(0, 4096, 4096) == -1
    assert ws_bytes(32, 4096, 4100) == -1            # K not a multiple of the 32-deep k-step
    assert ws_bytes(32, 4096, 128) == -1             # too shallow to cut
    assert ws_bytes(32, 8, 4096) == -1               # less than one column tile
    assert ws_bytes(32, 16384 * 16 + 16, 4096) == -1  # more column tiles than ticket words


def test_workspace_layout_is_shape_independent_at_its_head():
    """Launches of different widths share one workspace: the ticket region has ONE size (a per-shape size put a narrow launch's
    partial sums where a wide launch keeps its tickets - found by the GPU tests of the first cut)."""
    assert ws_bytes(20, 1000, 512) > TICKETS and ws_bytes(64, 32000, 4096) > TICKETS
    assert (ws_bytes(64, 32000, 4096) - TICKETS) % 1024 == 0
