"""Minimal multi-model PDB writer replacing ``mdtraj.Trajectory(...).save_pdb`` in
sample.py:244-247 / utils.py:215-218 (coordinates arrive in nm = Angstrom / 10 there and mdtraj
writes Angstrom, so the file holds the sampled Angstrom coordinates)."""
from __future__ import annotations

import numpy as np

from . import specs


def save_pdb(path: str, coords_angstrom, mol: str) -> None:
    xyz = np.asarray(coords_angstrom, dtype=np.float64)
    if xyz.ndim != 3 or xyz.shape[-1] != 3:
        raise ValueError("coords must be (frames, beads, 3)")
    atoms = specs.pdb_atoms(mol)
    if len(atoms) != xyz.shape[1]:
        raise ValueError("topology and coordinates disagree on the number of beads")
    with open(path, "w") as f:
        f.write("REMARK   1 CREATED WITH two-for-one-diffusion_amd\n")
        for m, frame in enumerate(xyz):
            f.write(f"MODEL     {m:4d}\n")
            for i, ((name, res, resid, elem), (x, y, z)) in enumerate(zip(atoms, frame)):
                aname = f" {name:<3s}" if len(name) < 4 else name
                f.write(f"ATOM  {i + 1:5d} {aname} {res:>3s} A{resid:4d}    {x:8.3f}{y:8.3f}{z:8.3f}"
                        f"  1.00  0.00          {elem:>2s}  \n")
            f.write(f"TER   {len(atoms) + 1:5d}      {atoms[-1][1]:>3s} A{atoms[-1][2]:4d}\n")
            f.write("ENDMDL\n")
        f.write("END\n")
