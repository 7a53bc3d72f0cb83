"""The C++ header shims (merlin-zkevm-prover_amd/host/*.hpp) are the host-side mirror of the reference's
interface: this builds a C++ program that uses them exactly like src/starkpil does and runs it on the GPU."""
import os, subprocess
import pytest
import glo

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_starkpil_flow.cpp")
EXE = os.path.join(ROOT, "tests", "cpp", "test_starkpil_flow")


def build_exe():
    glo.build()
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "merlin-zkevm-prover_amd", "host"), SRC, "-o", EXE,
           "-L", os.path.join(ROOT, "merlin-zkevm-prover_amd"), "-lmi_stark", "-L", os.path.join(ROOT, "oracle"), "-lgl_oracle",
           "-Wl,-rpath," + os.path.join(ROOT, "merlin-zkevm-prover_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"),
           "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    subprocess.check_call(cmd)


def test_shims_compile_against_the_c_abi():
    """CPU-side: the reference-style C++ code compiles and links against libmi_stark.so (no GPU needed)."""
    build_exe()
    assert os.path.exists(EXE)


@pytest.mark.gpu
def test_starkpil_flow_on_gpu():
    build_exe()
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL OK" in r.stdout
