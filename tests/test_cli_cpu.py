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
