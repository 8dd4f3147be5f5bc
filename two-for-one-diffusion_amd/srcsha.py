"""Source hash of the device code: sha256 over csrc/* (sorted by name) + include/dff.h, first 16 hex digits.

build.sh computes the same value and compiles it into the library (`dff_version()` ends in ``src=<hash>``);
tools_profile_report.py stamps it into every profiles/**/traffic.json; bench.py reports the rocprofv3 counters of a
traffic.json only when its hash is the running library's (otherwise ``roofline.profile_stale`` = true)."""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def tree_sha(root: str = None) -> str:
    """Hash of the sources in the tree (what build.sh would compile into the library right now)."""
    root = root or os.path.dirname(_HERE)
    csrc = os.path.join(root, os.path.basename(_HERE), "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(csrc)):   # byte order, as `LC_ALL=C sort` in build.sh
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    with open(os.path.join(root, "include", "dff.h"), "rb") as f:
        h.update(f.read())
    return h.hexdigest()[:16]


def library_sha(lib=None) -> str:
    """Hash the LOADED library was built from (parsed out of dff_version()); "unknown" for a build without it."""
    if lib is None:
        from .binding import load_library
        lib = load_library()
    v = lib.dff_version().decode()
    return v.rsplit("src=", 1)[1].strip() if "src=" in v else "unknown"
