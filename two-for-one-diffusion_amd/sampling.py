"""Batch orchestration (evaluate/evaluators.py:874-901, utils.py:201-212) and the multi-GPU
sharding that replaces the reference's ``nn.DataParallel`` (sample.py:180-190,204-214).

Samples / trajectories are independent units: rank r of W takes a contiguous share, draws from
its own Philox sub-stream (the global sample / trajectory index is the Philox counter, so results
do not depend on W), and the only collective is one ``all_gather`` of the finished result.
"""
from __future__ import annotations

import os
from typing import List, Tuple

import torch


def num_to_groups(num: int, divisor: int) -> List[int]:
    """evaluate/evaluators.py:891-901."""
    groups, remainder = num // divisor, num % divisor
    arr = [divisor] * groups
    if remainder > 0:
        arr.append(remainder)
    return arr


class SamplerWrapper:
    """utils.py:201-212: ``sampler(batch_size=b) -> model.sample(batch_size=b)``."""

    def __init__(self, model):
        self.model = model

    def eval(self):
        return self

    def to(self, *_a, **_k):
        return self

    def __call__(self, **kwargs):
        return self.model.sample(**kwargs)


def sample_from_model(sampler, num_saved_samples: int, batch_size: int, verbose: bool = False, to_cpu: bool = True):
    """evaluate/evaluators.py:874-888: returns a CPU tensor (num_saved_samples, N, 3).  to_cpu=False keeps the result on the
    device (the Langevin initial structures are consumed there: no PCIe round trip at config 4's 2048 x 8 simulations)."""
    print(f"Generating {num_saved_samples} samples per GPU. This may take some time.")
    batches = num_to_groups(num_saved_samples, batch_size)
    all_mol_list = []
    for i, bs in enumerate(batches):
        all_mol_list.append(sampler(batch_size=bs))
        if verbose:
            print(f"Batch {i + 1} from {len(batches)} generated")
    all_mol = torch.cat(all_mol_list, dim=0)
    if to_cpu:
        all_mol = all_mol.cpu()
    print(f"{len(all_mol)} samples generated")
    return all_mol


# ---------------------------------------------------------------------------- sharding
def shard_range(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) share of ``total`` independent units for ``rank`` of ``world``;
    the first ``total % world`` ranks take one extra unit."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


_warned_knobs = False


def dist_env() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (1-process default).

    TEST KNOBS, honoured only together with DFF_TEST_KNOBS=1 (a variable leaked from a test shell must not silently put
    every rank of a production job on one GPU): DFF_DEVICE overrides the device index (several ranks on ONE GPU: how the
    multi-rank path is exercised on a one-GPU box) and DFF_DIST_BACKEND the process-group backend (`dist_backend`).
    Rank 0 says so on stderr when they are in effect; without DFF_TEST_KNOBS they are ignored, also with a note."""
    global _warned_knobs
    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    knobs = {k: os.environ[k] for k in ("DFF_DEVICE", "DFF_DIST_BACKEND") if k in os.environ}
    if knobs:
        on = os.environ.get("DFF_TEST_KNOBS") == "1"
        if on and "DFF_DEVICE" in knobs:
            local = int(knobs["DFF_DEVICE"])
            # several ranks on one GPU: the two-workgroups-per-protein kernels need a GPU to themselves (both blocks of a pair
            # resident at once) -- off for models created from here on (dff_model_create reads DFF_PAIR)
            if int(os.environ.get("WORLD_SIZE", 1)) > 1:
                os.environ.setdefault("DFF_PAIR", "0")
        if rank == 0 and not _warned_knobs:
            import sys
            print(f"dff_amd: test knobs {knobs} are {'IN EFFECT (DFF_TEST_KNOBS=1)' if on else 'IGNORED (set DFF_TEST_KNOBS=1 to use them)'}",
                  file=sys.stderr)
            _warned_knobs = True
    return (rank, local, int(os.environ.get("WORLD_SIZE", 1)))


def dist_backend() -> str:
    """Process-group backend: "nccl" (= RCCL over xGMI) unless the test knob DFF_DIST_BACKEND is in effect."""
    if os.environ.get("DFF_TEST_KNOBS") == "1" and os.environ.get("DFF_DIST_BACKEND"):
        return os.environ["DFF_DIST_BACKEND"]
    return "nccl"


def gather_variable(local: torch.Tensor, total: int, world: int, group=None) -> torch.Tensor:
    """all_gather of per-rank result blocks whose leading sizes follow ``shard_range``.

    One collective (RCCL over xGMI on the GPU box, gloo in CPU tests): blocks are padded to the
    largest share, gathered, and trimmed.  Returns the (total, ...) tensor on every rank."""
    import torch.distributed as dist
    if world == 1:
        return local
    sizes = [shard_range(total, r, world) for r in range(world)]
    mx = max(b - a for a, b in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(out, pad, group=group)
    return torch.cat([o[: b - a] for o, (a, b) in zip(out, sizes)], dim=0)
