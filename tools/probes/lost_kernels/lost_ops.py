"""Laboratory front-ends of the two kernels taken out of libvdd_hip.so in round 6 (measured slower than what ships; DESIGN.md
section 6): the 17 - 64-row slab projections (vdd_skinny_slab.hip) and the persistent few-row decode layers (vdd_layer_persistent.hip).
`python tools/probes/lost_kernels/lost_ops.py` (or build_lost_lib()) compiles libvdd_lost.so next to this file; the package, bench.py
and the default test suite never load it.  Tests: `pytest tools/probes/lost_kernels -m probe` on a GPU box (they are marked `probe`
AND `gpu`)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

from llava_align_amd import _build, _lib
from llava_align_amd.ops import _I, _L, _F, _P, _MODEL_DT, _dt, _st

LIB = os.path.join(HERE, "libvdd_lost.so")
_SRCS = ("vdd_skinny_slab.hip", "vdd_layer_persistent.hip")


def build_lost_lib(force: bool = False, probe: bool = False) -> str:
    """hipcc for gfx950, the product's flags; `probe`: -DVDD_PROBE_BUILD (per-wave timeline stamps, tools/slab_timeline.py)."""
    srcs = [os.path.join(HERE, f) for f in _SRCS + ("vdd_lost_dispatch.hip", "vdd_lost.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(s) <= os.path.getmtime(LIB) for s in srcs):
        return LIB
    obj = os.path.join(ROOT, "build", "obj_lost")
    os.makedirs(obj, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", _build.CSRC, "-I", HERE]
    units = [(os.path.join(HERE, f), os.path.join(obj, f"{f[:-4]}.{tag}.o"), [f"-DVDD_ELEM={val}"]) for f in _SRCS for tag, val in _build.ELEMS]
    units.append((os.path.join(HERE, "vdd_lost_dispatch.hip"), os.path.join(obj, "vdd_lost_dispatch.o"), []))
    extra = ["-DVDD_PROBE_BUILD"] if probe else []
    run = lambda u: subprocess.run([_build.hipcc(), *_build.CFLAGS, *extra, *u[2], *inc, "-c", u[0], "-o", u[1]], check=True)
    with ThreadPoolExecutor(max_workers=len(units)) as ex:
        list(ex.map(run, units))
    subprocess.run([_build.hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[u[1] for u in units], "-o", LIB], check=True)
    return LIB


def __getattr__(name):            # everything else (linear, rmsnorm, ...) is the product's: the lab tests compare against it
    from llava_align_amd import ops as product
    return getattr(product, name)


_lost = None


def _lib_ready():
    global _lost
    if _lost is None:
        if not os.path.exists(LIB):
            raise RuntimeError(f"{LIB} is missing: build it with `python tools/probes/lost_kernels/lost_ops.py` (laboratory code, not part of the package)")
        _lost = C.CDLL(LIB)
        _lost.vdd_skinny_slab.argtypes = [_P, _P, _I, _P, _F, _P, _P, _P, _P, _I, _I, _I, _L, _L, _L, _I, _P, _L, _I, _P]
        _lost.vdd_decode_layers.argtypes = [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _L, _I, _L, _I, _I, _P, _L, _I, _P]
        _lost.vdd_skinny_slab.restype = _lost.vdd_decode_layers.restype = C.c_int
        _lost.vdd_skinny_slab_status_offset.argtypes, _lost.vdd_skinny_slab_status_offset.restype = [], C.c_int64
    return _lost


# ---------------------------------------------------------------- 17 - 64 rows: K cut over workgroups (vdd_skinny_slab.hip)
SLAB_MIN_M, SLAB_MAX_M = 17, 64     # rows the slab projections take in the decoder layer (below: the <= 16-row kernels above)
_slab_ws = {}
_slab_retired = []


def _slab_workspace(device, nbytes):
    """Tile tickets + fp32 partial slabs of the slab projections, one per (device, STREAM) like the GEMM's scratch: launches on one
    stream are ordered and share it; the tickets start at zero and every completed launch leaves them zero."""
    key = (torch.device(device).index, torch.cuda.current_stream(device).cuda_stream)
    ws = _slab_ws.get(key)
    if ws is None or ws.numel() < nbytes:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the slab workspace must exist before a graph capture (run the step once eagerly on this stream)")
        if ws is not None:
            _slab_retired.append(ws)              # a captured graph may still point at it: never freed while the process lives
        ws = _slab_ws[key] = torch.zeros(max(int(nbytes), 48 << 20), dtype=torch.uint8, device=device)
    return ws


def slab_workspace_reset():
    _slab_ws.clear()


def slab_status(workspace=None, device=None) -> int:
    """The give-up word of the slab launches on this workspace (forces a sync): 0, or 1 + the index of the first workgroup whose bounded
    wait for its team timed out - the outputs of that launch are garbage (a grid that was not co-resident: CU mask, a concurrent
    kernel, a partitioned part).  slab_linear(check=True) raises on it."""
    if workspace is None:
        dev = torch.device(device if device is not None else "cuda")
        workspace = _slab_ws.get((dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream))
        if workspace is None:
            return 0
    off = int(_lib_ready().vdd_skinny_slab_status_offset())
    return int(workspace[off:off + 4].view(torch.int32).item())


def slab_serves(M, N, K, swiglu=False) -> bool:
    lib = _lib_ready()
    lib.vdd_skinny_slab_workspace_bytes.argtypes, lib.vdd_skinny_slab_workspace_bytes.restype = [_I, _I, _I, _I], C.c_int64
    return lib.vdd_skinny_slab_workspace_bytes(M, N, K, int(swiglu)) >= 0


def slab_linear(x, w, resid=None, ss=None, ln_w=None, eps=0.0, swiglu=False, want_ss=False, out=None, workspace=None, check=False):
    """x [M <= 64, K] @ w[N, K]^T on the slab kernels.  ss / ln_w: x is the un-normalised residual stream, normalised as it is staged
    (ss [M, nss] partial sums of squares from a previous call with want_ss).  resid: y = rnd(rnd(x w^T) + resid).  swiglu: w = [Wg; Wu],
    y = silu(x Wg^T) * (x Wu^T).  want_ss: also returns ss_out [M, ceil(N / 16)].  A captured step passes its own `workspace`."""
    dt = _dt(x, w, resid, ln_w)
    M, K = x.shape
    N = w.shape[0] // 2 if swiglu else w.shape[0]
    lib = _lib_ready()
    lib.vdd_skinny_slab_workspace_bytes.argtypes, lib.vdd_skinny_slab_workspace_bytes.restype = [_I, _I, _I, _I], C.c_int64
    need = lib.vdd_skinny_slab_workspace_bytes(M, N, K, int(swiglu))
    if need < 0:
        raise ValueError(f"slab_linear: shape M={M} N={N} K={K} is not served")
    ws = _slab_workspace(x.device, need) if workspace is None else workspace
    out = torch.empty(M, N, dtype=x.dtype, device=x.device) if out is None else out
    ss_out = torch.empty(M, (N + 15) // 16, dtype=torch.float32, device=x.device) if want_ss else None
    _lib.check(lib.vdd_skinny_slab(x.data_ptr(), ss.data_ptr() if ss is not None else None, ss.shape[1] if ss is not None else 0,
                                   ln_w.data_ptr() if ss is not None else None, eps, w.data_ptr(), resid.data_ptr() if resid is not None else None,
                                   out.data_ptr(), ss_out.data_ptr() if want_ss else None, M, N, K, x.stride(0),
                                   resid.stride(0) if resid is not None else 0, out.stride(0), int(swiglu), ws.data_ptr(), ws.numel(), dt, _st(x)))
    if check and slab_status(ws):
        code = slab_status(ws)
        ws.zero_()
        raise RuntimeError(f"slab projection gave up: workgroup {code - 1} waited for its team past the bound (grid not co-resident); workspace re-zeroed")
    return (out, ss_out) if want_ss else out


# ---------------------------------------------------------------- persistent few-row decode layers (vdd_layer_persistent.hip)
PERSISTENT_LAYERS = True     # one launch for ALL decoder layers of a 1 - 4 row decode step (False: the five-launch layer)
LAYER_DESC_FIELDS = 11       # vdd_layer_desc: ln1, wqkv, bqkv, wo, ln2, wgu, wd, k_own, v_own, k_pre, v_pre


def decode_layers_max_rows(d: int, H: int, Hkv: int, F: int, D: int, n_layers: int, dtype=torch.bfloat16) -> int:
    """Rows one persistent launch takes for this model shape on the current device (0: not served - GQA, head_dim != 128, ...)."""
    if not PERSISTENT_LAYERS or Hkv != H or n_layers < 1 or n_layers > 120:
        return 0
    lib = _lib_ready()
    lib.vdd_decode_layers_max_rows.argtypes, lib.vdd_decode_layers_max_rows.restype = [_I, _I, _I, _I, _I, _I], C.c_int
    return int(lib.vdd_decode_layers_max_rows(d, H, F, D, n_layers, _MODEL_DT[dtype]))


def decode_layers_workspace(M: int, d: int, H: int, F: int, D: int, device, dtype=torch.bfloat16) -> torch.Tensor:
    """Granule exchange buffers + launch counter + give-up word of the persistent layers: ZEROED once, then owned by the launches
    (the epochs of the hand-offs derive from the launch counter in it, also under graph replay).  Allocate before a capture."""
    lib = _lib_ready()
    lib.vdd_decode_layers_workspace_bytes.restype = C.c_int64
    lib.vdd_decode_layers_workspace_bytes.argtypes = [_I, _I, _I, _I, _I, _I]
    n = int(lib.vdd_decode_layers_workspace_bytes(M, d, H, F, D, _MODEL_DT[dtype]))
    if n <= 0:
        raise ValueError(f"persistent decode layers do not serve M={M}, d={d}, H={H}, F={F}, D={D}")
    return torch.zeros(n, dtype=torch.uint8, device=device)


def layer_descriptors(layers, device) -> torch.Tensor:
    """[n_layers, 11] int64 device tensor of vdd_layer_desc records; `layers`: per layer a dict with the tensors ln1, wqkv, bqkv (or
    None), wo, ln2, wgu, wd, k_own, v_own, k_pre, v_pre.  The caller keeps the tensors alive."""
    keys = ("ln1", "wqkv", "bqkv", "wo", "ln2", "wgu", "wd", "k_own", "v_own", "k_pre", "v_pre")
    rows = [[0 if l.get(k) is None else l[k].data_ptr() for k in keys] for l in layers]
    return torch.tensor(rows, dtype=torch.int64).to(device)


def decode_layers(desc, n_layers, resid_in, pos, cpos, slot, cos_sin, rows, H, F, D, eps, slot_stride, t_max, prefix_stride, prefix_tmax,
                  has_qkv_bias, workspace, resid_out=None, ss_out=None):
    """All decoder layers of one decode step for M <= decode_layers_max_rows rows in ONE persistent launch.  resid_in [M, d]: the
    embeddings.  Returns (resid [M, d], ss [M, n]): the residual stream behind the last layer + partial sums of squares of its rows,
    what linear_normed takes for the final norm + lm_head.  The KV pools behind `desc` get the new token's K / V."""
    dt = _dt(resid_in)
    M, d = resid_in.shape
    resid_out = torch.empty_like(resid_in) if resid_out is None else resid_out
    if ss_out is None:
        lib = _lib_ready()
        lib.vdd_decode_layers_ss_cols.argtypes, lib.vdd_decode_layers_ss_cols.restype = [_I, _I, _I, _I, _I], C.c_int
        ss_out = torch.empty(M, int(lib.vdd_decode_layers_ss_cols(d, H, F, D, dt)), dtype=torch.float32, device=resid_in.device)
    _lib.check(_lib_ready().vdd_decode_layers(desc.data_ptr(), n_layers, resid_in.data_ptr(), resid_out.data_ptr(), ss_out.data_ptr(),
                                              pos.data_ptr(), cpos.data_ptr(), slot.data_ptr(), cos_sin.data_ptr(), rows.data_ptr(), M, d, H, H,
                                              F, D, eps, D ** -0.5, slot_stride, t_max, prefix_stride, prefix_tmax, 1 if has_qkv_bias else 0,
                                              workspace.data_ptr(), workspace.numel(), dt, _st(resid_in)))
    return resid_out, ss_out


def decode_layers_status(workspace) -> int:
    """The give-up word of the persistent layers (forces a sync): 0, or the code of the first wait that timed out
    (low 16 bits: phase code of vdd_layer_persistent.hip, high bits: workgroup) - the step's results are garbage then."""
    return int(workspace[4:8].view(torch.int32).item())


if __name__ == "__main__":
    print(build_lost_lib(force=True, probe="--probe" in sys.argv))
