"""CPU: host logic of the `clustering density` command line that needs no GPU -- option parsing,
help, error exits, and that the binary refuses to run without a HIP device (no CPU fallback)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "clustering_amd", "bin", "clustering")


def run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=60)


def test_binary_is_built():
    assert os.path.exists(CLI), "run __graft_entry__.build() first"


def test_general_help_and_modes():
    r = run()
    assert r.returncode != 0 and "clustering density -h" in r.stderr
    r = run("mpp", "-h")
    assert r.returncode != 0 and "unrecognized mode 'mpp'" in r.stderr


def test_density_help_lists_reference_options():
    r = run("density", "-h")
    assert r.returncode == 0
    for opt in ("--file", "--radius", "--radii", "--population", "--free-energy",
                "--free-energy-input", "--nearest-neighbors", "--nearest-neighbors-input",
                "--nthreads", "--verbose"):
        assert opt in r.stdout, opt


def test_missing_required_file_option():
    r = run("density", "-r", "0.2")
    assert r.returncode != 0 and "'--file' is required" in r.stderr


def test_unknown_option():
    r = run("density", "-f", "x", "--frobnicate")
    assert r.returncode != 0 and "unrecognised option" in r.stderr


def test_no_gpu_means_exit_not_fallback(tmp_path):
    import torch
    if torch.cuda.is_available():
        return
    p = tmp_path / "coords.txt"
    p.write_text("0.0 1.0\n1.0 0.0\n")
    r = run("density", "-f", str(p), "-r", "0.5", "-p", str(tmp_path / "pop"))
    assert r.returncode != 0
    assert "no HIP-compatible GPUs found" in r.stderr
    assert not (tmp_path / "pop").exists()


def test_shim_header_is_legal_next_to_the_reference_headers(tmp_path):
    """INTEGRATION.md section 2 claims density_clustering_hip.hpp can be included next to the reference's
    own headers (it re-declares Tools::Neighbor / Neighborhood and the CUDA:: entry points).  Syntax check
    only, in the build container only: tools.hpp is read where it lies under /root/reference, its
    cmake-generated config.hpp is produced from the reference's own template the way configure_file does
    (CMakeLists.txt:57,88); nothing of the reference is copied into the repo or travels to the GPU box.
    (density_clustering_common.hpp cannot take part: it pulls in Boost, which this image lacks.)"""
    import pytest
    ref = "/root/reference"
    if not os.path.exists(os.path.join(ref, "src", "tools.hpp")):
        pytest.skip("reference sources are only present in the build container")
    template = open(os.path.join(ref, "config.hpp.cmake.in")).read()
    (tmp_path / "config.hpp").write_text(template.replace("@DC_MEM_ALIGNMENT@", "32"))
    shim = os.path.join(os.path.dirname(CLI), "..", "csrc", "density_clustering_hip.hpp")
    (tmp_path / "tu.cpp").write_text(
        '#include "tools.hpp"\n#include "%s"\n'
        "static_assert(std::is_same<Clustering::Density::CUDA::Neighborhood, Clustering::Tools::Neighborhood>::value, \"\");\n"
        "namespace Clustering { namespace Density { typedef std::map<float, std::vector<std::size_t>> Pops; } }\n"
        "int main() { return 0; }\n" % os.path.abspath(shim))
    # (-include limits: tools.hxx:244 relies on a transitive include that GCC 11 no longer provides)
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-include", "limits", "-I", str(tmp_path), "-I", os.path.join(ref, "src"),
                        str(tmp_path / "tu.cpp")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_malformed_coordinate_files_are_refused(tmp_path):
    """The reference counts non-empty lines as rows and then reads rows x columns numbers with `ifs >> float`
    (tools.hxx:52-108): a ragged file, a comment or a token iostream does not take ('nan', 'inf', hex) makes
    it read garbage.  This reader refuses such files before any GPU work (so the check runs without a GPU)."""
    cases = {"ragged": "0 1\n2 3 4\n5 6\n", "comment": "# x y\n0 1\n2 3\n", "nan": "0 1\nnan 3\n",
             "inf": "0 1\n2 inf\n", "hex": "0 1\n0x1p3 3\n", "short": "0 1\n2\n"}
    for name, text in cases.items():
        f = tmp_path / name
        f.write_text(text)
        r = run("density", "-f", str(f), "-r", "1", "-p", str(tmp_path / "pop"))
        assert r.returncode != 0, name
        assert ("readable numbers" in r.stderr) or ("no HIP" in r.stderr) or ("GPU" in r.stderr), (name, r.stderr)
