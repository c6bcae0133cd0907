"""tools/lds_bank_model.py (MI355X_MICROARCH.md's LDS banking rules applied to the update kernels' piece-image accesses): every row / transposed READ of the chains, the dW2
product, the dz1 stage and the weight image must stay conflict-free; the 8-byte piece stores and `pair_load_pieces2` are the documented 2-way cases (profiles/README.md)."""
import re, subprocess, sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_fragment_reads_are_conflict_free_and_piece_stores_two_way():
    out = subprocess.run([sys.executable, str(ROOT / "tools" / "lds_bank_model.py")], capture_output=True, text=True, check=True).stdout
    rows = [m.groups() for m in (re.match(r"(.+?)\s+(\d+) cycles \(conflict-free (\d+)\)", l) for l in out.splitlines()) if m]
    assert len(rows) >= 80
    for name, cyc, ideal in rows:
        cyc, ideal = int(cyc), int(ideal)
        if "ds_write_b64" in name or "pair_load_pieces2" in name:
            assert cyc == 2 * ideal, (name, cyc, ideal)
        else:
            assert cyc == ideal, (name, cyc, ideal)
