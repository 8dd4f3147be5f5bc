"""two-for-one-diffusion_amd -- MI355X-native sampler for the denoising force field of
microsoft/two-for-one-diffusion: the score network (energy forward + hand-written VJP) inside
the DDPM reverse loop and the Langevin integrator, as one persistent HIP kernel behind a C ABI
(include/dff.h, libdff_amd.so).  Host side mirrors the reference's Python interfaces.

The directory name is not a Python identifier; import it as ``import dff_amd`` (alias module at
the repository root) or ``importlib.import_module("two-for-one-diffusion_amd")``.
"""
from . import binding, specs, weights  # noqa: F401
from .binding import DffLibraryError, Model, load_library  # noqa: F401
from .sampling import SamplerWrapper, num_to_groups, sample_from_model  # noqa: F401

__all__ = ["binding", "specs", "weights", "DffLibraryError", "Model", "load_library", "SamplerWrapper",
           "num_to_groups", "sample_from_model", "GraphTransformer", "GaussianDiffusion",
           "LangevinDiffusion", "ForcesWrapper"]


def __getattr__(name):  # torch-dependent pieces are imported on first use
    if name == "GraphTransformer":
        from .score import GraphTransformer
        return GraphTransformer
    if name == "GaussianDiffusion":
        from .ddpm import GaussianDiffusion
        return GaussianDiffusion
    if name in ("LangevinDiffusion", "ForcesWrapper"):
        from . import langevin
        return getattr(langevin, name)
    raise AttributeError(name)
