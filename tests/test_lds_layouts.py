"""The LDS images of k_tile2g's leaf blocks (G2Layout<M, PB>, dct_pruning_amd/csrc/tile2g.hip) are bank-conflict-free for every
row and column access of a pass: re-checked on the CPU by the search tool's model (32 banks, 32 lanes per cycle, idle lanes
broadcasting the nearest active lane's address)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_tile2g_layouts_are_conflict_free():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "g2_layout_search.py")], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=120)
    assert p.returncode == 0, p.stdout
    assert p.stdout.count("conflict-free") >= 6 and "CONFLICTS" not in p.stdout
