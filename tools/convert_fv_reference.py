#!/usr/bin/env python3
"""Re-encode the reference's stored FV solutions (data/validation/fv*/Re*/solution.vts, 1.2 MB of
zlib+base64 VTK XML each, with six arrays) as compact ``solution.npz`` fixtures holding only what
``compute_validation_errors`` consumes: point coordinates and u, v.  Data, not code; run once here."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "02689-advancednumericalalgorithmp3_amd" / "src"))
from solvers.vtkio import read_vts  # noqa: E402

SRC = Path("/root/reference/data/validation")
DST = ROOT / "02689-advancednumericalalgorithmp3_amd" / "data" / "validation"
for vts in sorted(SRC.glob("fv*/Re*/solution.vts")):
    g = read_vts(vts)
    out = DST / vts.parent.parent.name / vts.parent.name / "solution.npz"
    out.parent.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(out, x=g["points"][:, 0], y=g["points"][:, 1], u=g["point_data"]["u"],
                        v=g["point_data"]["v"], Re=g["field_data"].get("Re", np.array([0])))
    print(out.relative_to(ROOT), out.stat().st_size // 1024, "KB")
