"""The arithmetic forms of the RFC 1951 length / distance code fields (csrc/zh_tables.h, used on
the device hot paths) equal the generated tables for every length 3..258 and distance 1..32768."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_arithmetic_code_fields_match_the_tables(tmp_path):
    exe = str(tmp_path / "tabcheck")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "zippy_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tests", "native", "tabcheck.cpp")], check=True)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
