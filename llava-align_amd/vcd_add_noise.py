"""Drop-in for the reference's `vcd_utils/vcd_add_noise.py`: add_diffusion_noise(image, t).

The 1000-step sigmoid schedule (vcd_add_noise.py:7-16) is tabulated once per process
with the same fp32 torch ops the reference re-runs on every call; the noising itself
(:18-22) is a HIP kernel (`vdd_add_diffusion_noise`) with an in-kernel Philox/Box-Muller
normal generator seeded from torch's global seed.  A CPU image tensor (what the
reference's LLaVA drivers pass, llava_calibrate.py:152) is moved to the GPU for the
kernel and returned on its original device; without a GPU this raises — no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib

_schedule = None
_DT = {torch.float32: _lib.VDD_F32, torch.float16: _lib.VDD_F16, torch.bfloat16: _lib.VDD_BF16}


_calls: dict = {}


def reset_noise_calls(seed: Optional[int] = None):
    """Restart the per-seed call counter (all seeds when None): the next add_diffusion_noise(..., seed=s) draws call 0 again."""
    if seed is None:
        _calls.clear()
    else:
        _calls.pop(int(seed) & 0xFFFFFFFFFFFFFFFF, None)


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def schedule(num_steps: int = 1000):
    """(sqrt(abar), sqrt(1-abar)) as python floats per step — vcd_add_noise.py:7-16."""
    global _schedule
    if _schedule is None or len(_schedule[0]) != num_steps:
        betas = torch.sigmoid(torch.linspace(-6, 6, num_steps)) * (0.5e-2 - 1e-5) + 1e-5
        abar = torch.cumprod(1 - betas, dim=0)
        _schedule = (torch.sqrt(abar).tolist(), torch.sqrt(1 - abar).tolist())
    return _schedule


def add_diffusion_noise(image_tensor: torch.Tensor, noise_step: int, noise: Optional[torch.Tensor] = None,
                        seed=None) -> torch.Tensor:
    """vcd_add_noise.py:3-28.  `noise`: explicit epsilon (tests).  Random numbers, by `seed`:
      None             like the reference: every call draws fresh noise from torch's global generator state (torch.manual_seed
                       reproduces a run);
      (seed, call)     a PURE function of (seed, call, shape): `call` is the caller's own index of this draw (the question number in a
                       driver loop) - thread-safe, repeatable in-process, independent of what else was noised;
      int              shorthand for (seed, n) with n = the number of earlier int-seeded calls under that seed in this PROCESS
                       (module state, not thread-safe: consecutive images get different epsilon; reset_noise_calls() restarts it)."""
    lib = _lib.load_lib()
    if not torch.cuda.is_available():
        raise _lib.VddLibraryError("add_diffusion_noise needs a GPU: this package has no CPU path")
    if image_tensor.dtype not in _DT:
        raise ValueError(f"unsupported image dtype {image_tensor.dtype}")
    a, b = schedule()
    t = int(noise_step)
    src_dev = image_tensor.device
    x = image_tensor.to("cuda", non_blocking=True).contiguous()
    y = torch.empty_like(x)
    eps_ptr = None
    if noise is not None:
        eps = noise.to(device=x.device, dtype=torch.float32).contiguous()
        if eps.numel() != x.numel():
            raise ValueError("noise must have as many elements as the image")
        eps_ptr = eps.data_ptr()
    lib.vdd_add_diffusion_noise.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float,
                                            C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    lib.vdd_add_diffusion_noise.restype = C.c_int
    call_idx = None
    if isinstance(seed, (tuple, list)):
        seed, call_idx = int(seed[0]), int(seed[1])
        if call_idx < 0:
            raise ValueError("add_diffusion_noise: seed=(seed, call) takes a non-negative call index")
    sd = (torch.initial_seed() if seed is None else int(seed)) & 0xFFFFFFFFFFFFFFFF
    # Philox counter of element i = off + i / 4: the low 24 bits index inside the call (images up to 2^26 elements), the upper 40
    # the CALL.  No seed: the call id is drawn from torch's generator (each call advances it like the reference's randn_like,
    # vcd_add_noise.py:24; torch.manual_seed reproduces a run; 40 bits: two of ~10^3 images collide with probability ~5e-7).
    # seed=(s, call): the caller names the call.  Plain int seed: the call id counts the calls made under that seed in this process, so
    # consecutive images get different epsilon (reset_noise_calls() restarts it).  The data-parallel rank is folded into
    # the key so that ranks seeded alike do not noise their shards with one stream.
    from .sampling import fresh_offset
    if x.numel() > (1 << 26):
        raise ValueError("add_diffusion_noise: more than 2^26 elements in one call")
    if seed is None:
        call = fresh_offset() & ((1 << 40) - 1)
    elif call_idx is not None:
        call = call_idx
    else:
        call = _calls[sd] = _calls.get(sd, -1) + 1
    off = (call & ((1 << 40) - 1)) << 24
    sd = (sd ^ (_rank() * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    with torch.cuda.device(x.device):
        _lib.check(lib.vdd_add_diffusion_noise(x.data_ptr(), y.data_ptr(), x.numel(), _DT[x.dtype], a[t], b[t],
                                               eps_ptr, sd, off, torch.cuda.current_stream(x.device).cuda_stream))
    return y if src_dev.type == "cuda" else y.to(src_dev)
