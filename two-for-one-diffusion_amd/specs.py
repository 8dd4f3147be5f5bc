"""Static molecule metadata the samplers need (the reference gets it from mdtraj / its datasets).

Sources: bead counts = CA records of datasets/folded_pdbs/*.pdb and
datasets/dataset_utils_empty.py:203-221 (alanine dipeptide: 5); norm_std =
datasets/dataset_utils_empty.py:38-48; temperatures = dynamics/langevin.py:11-26; default masses
= sample.py:216-221.  Only the empty-dataset branch (``data_folder=None``) is covered, which is
all ``sample.py`` needs: ``num_beads``, ``bead_onehot``, ``std``, ``topology``.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional


@dataclass(frozen=True)
class Molecule:
    key: str            # canonical lower-case name
    mol: str            # args.mol as stored in args.pickle
    n_beads: int
    temperature: float  # K, temp_dict[mol.upper()]
    default_mass: float
    sequence: Optional[str]  # one-letter CA sequence for the PDB writer (None: generic)


_AA3 = dict(A="ALA", R="ARG", N="ASN", D="ASP", C="CYS", Q="GLN", E="GLU", G="GLY", H="HIS", I="ILE",
            L="LEU", K="LYS", M="MET", F="PHE", P="PRO", S="SER", T="THR", W="TRP", Y="TYR", V="VAL",
            X="NLE")

MOLECULES = {
    "alanine_dipeptide_fuberlin": Molecule("alanine_dipeptide_fuberlin", "alanine_dipeptide_fuberlin", 5, 300, 12.8, None),
    "alanine_dipeptide_mdshare": Molecule("alanine_dipeptide_mdshare", "alanine_dipeptide_mdshare", 5, 300, 12.8, None),
    "chignolin": Molecule("chignolin", "CHIGNOLIN", 10, 340, 12.0, "YYDPETGTWY"),
    "trp_cage": Molecule("trp_cage", "TRP_CAGE", 20, 290, 12.0, "DAYAQWLADGGPSSGRPPPS"),
    "bba": Molecule("bba", "BBA", 28, 325, 12.0, "EQYTAKYKGRTFRNEKELRDFIEKFKGR"),
    "villin": Molecule("villin", "VILLIN", 35, 360, 12.0, "LSDEDFKAVFGMTRSAFANLPLWXQQHLXKEKGLF"),
    "protein_g": Molecule("protein_g", "PROTEIN_G", 56, 350, 12.0,
                          "DTYKLVIVLNGTTFTYTTEAVDAATAEKVFKQYANDAGVDGEWTYDAATKTFTVTE"),
}

# datasets/dataset_utils_empty.py:38-48 (alanine: one value per cross-validation fold)
NORM_STDS = {
    "chignolin": 3.113133430480957,
    "trp_cage": 5.08211088180542,
    "bba": 6.294918537139893,
    "villin": 6.082900047302246,
    "protein_g": 6.354289531707764,
    "alanine_fold1": 0.9449278712272644,
    "alanine_fold2": 0.944965124130249,
    "alanine_fold3": 0.9452606439590454,
    "alanine_fold4": 0.9454087018966675,
}


def lookup(mol: str) -> Molecule:
    k = mol.lower()
    if k not in MOLECULES:
        raise NotImplementedError(f"Invalid molecule name: {mol}")
    return MOLECULES[k]


def norm_std(mol: str, fold: int = 1) -> float:
    """CGDataset.std (datasets/dataset_utils_empty.py:196): per molecule; per fold for alanine."""
    k = mol.lower()
    if "alanine" in k:
        return NORM_STDS[f"alanine_fold{fold}"]
    return NORM_STDS[lookup(mol).key]


def default_masses(mol: str) -> List[float]:
    """sample.py:216-221."""
    m = lookup(mol)
    return [m.default_mass] * m.n_beads


# coarse-grained alanine dipeptide: the 5 heavy backbone atoms of datasets/folded_pdbs/ala2_cg.pdb
_ALA2_ATOMS = [("C", "ACE", 1, "C"), ("N", "ALA", 2, "N"), ("CA", "ALA", 2, "C"), ("C", "ALA", 2, "C"),
               ("N", "NME", 3, "N")]


def pdb_atoms(mol: str):
    """[(atom name, residue name, residue number, element)] per bead -- the topology
    ``CGDataset.topology`` carries in the reference (one CA per residue for the proteins,
    residue names as in datasets/folded_pdbs/*-0-c-alpha.pdb; 5 backbone atoms for ala2)."""
    m = lookup(mol)
    if m.sequence is None:
        return list(_ALA2_ATOMS)
    return [("CA", _AA3[c], i + 1, "C") for i, c in enumerate(m.sequence)]
