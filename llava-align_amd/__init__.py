"""MI355X-native visual contrastive decoding (VDD / VCD) for LLaVA-style VLMs.

Public surface mirrors the reference's (yfzhang114/LLaVA-Align):

    from llava_align_amd import evolve_vcd_sampling          # vcd_utils/vcd_sample.py:325
    from llava_align_amd import add_diffusion_noise          # vcd_utils/vcd_add_noise.py:3
    evolve_vcd_sampling()
    model.generate(ids, images=..., use_dd=True, use_dd_unk=True, cd_alpha=1, cd_beta=0.1, ...)

plus one line that puts the native engine (all-HIP model + fused sampling tail, in the model's own dtype) behind that same call on a
loaded LlavaLlamaForCausalLM (experiments/llava/model/builder.py:26-148):

    attach_engine(model)

The compute path is hand-written HIP for gfx950 behind a C ABI (include/vdd_hip.h,
libvdd_hip.so).  There is NO CPU fallback: anything that needs the library raises
VddLibraryError when it is missing.
"""
from ._lib import VddLibraryError, lib_path, load_lib  # noqa: F401
from .sampling import SampleOutput, WarpSpec, contrast_sample  # noqa: F401
from .vcd_sample import evolve_vcd_sampling, sample  # noqa: F401
from .vcd_add_noise import add_diffusion_noise  # noqa: F401
from .hf_adapter import attach_engine, detach_engine  # noqa: F401

__all__ = ["evolve_vcd_sampling", "sample", "contrast_sample", "WarpSpec", "SampleOutput",
           "add_diffusion_noise", "attach_engine", "detach_engine", "load_lib", "lib_path", "VddLibraryError"]
