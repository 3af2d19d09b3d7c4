"""CPU check of the flat/panel layout metadata and the reduction contract (no GPU): compiles
tests/cpp/layout_emulator.cpp against the product's flat_layout.cpp and runs it."""
import os
import subprocess

from conftest import ROOT


def test_layout_emulator(tmp_path):
    src = os.path.join(ROOT, "cuda-recommender_amd", "csrc")
    exe = str(tmp_path / "layout_emulator")
    subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", src, os.path.join(ROOT, "tests", "cpp", "layout_emulator.cpp"),
                    os.path.join(src, "flat_layout.cpp"), "-o", exe], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600).stdout
    assert "cases ok" in out, out
