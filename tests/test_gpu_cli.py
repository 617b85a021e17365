"""GPU: the `clustering density` command line and the C++ shim (reference container types) against
the oracle -- file formats as the reference's writers produce them (tools.cpp:42-56, 144-174)."""
import os
import subprocess

import numpy as np
import pytest

from clustering_amd.synth import gaussian_blobs

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "clustering_amd", "bin", "clustering")
SHIM = os.path.join(ROOT, "clustering_amd", "bin", "test_shim")


def data_lines(path):
    return [l for l in open(path).read().splitlines() if l and not l.startswith("#")]


def comments(path):
    out = {}
    for l in open(path).read().splitlines():
        if l.startswith("#@"):
            k, v = l[2:].split("=")
            out[k.strip()] = v.strip()
    return out


def write_coords(path, c):
    np.savetxt(path, c, fmt="%.9g")
    # what the reference's reader (and ours) sees after the decimal round trip
    return np.loadtxt(path, dtype=np.float64, ndmin=2).astype(np.float32)


def fmt_e(x):
    return "%e" % float(np.float32(x))


def fmt_g(x):
    return "%g" % float(np.float32(x))


def test_cli_single_radius_full_path(tmp_path, oracle):
    c = write_coords(tmp_path / "coords", gaussian_blobs(3000, 5, seed=42))
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-r", "0.1", "-p", str(tmp_path / "pop"),
                        "-d", str(tmp_path / "fe"), "-b", str(tmp_path / "nn"), "-v"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "using radius: 0.1" in r.stdout
    pops = oracle.populations(c, [0.1])[0]
    fe = oracle.free_energies(pops)
    nn = oracle.nearest_neighbors(c, fe)
    assert data_lines(tmp_path / "pop") == [str(int(p)) for p in pops]
    assert data_lines(tmp_path / "fe") == [fmt_e(v) for v in fe]
    want = ["%d %s %d %s" % (nn[0][i], fmt_g(nn[1][i]), nn[2][i], fmt_g(nn[3][i])) for i in range(len(c))]
    assert data_lines(tmp_path / "nn") == want
    cm = comments(tmp_path / "nn")
    assert cm["clustering_radius"] == "%.5f" % np.float32(0.1)
    assert cm["lumping_radius"] == "%.5f" % oracle.lumping_radius(oracle.sigma2(nn[1]))
    assert "-0.000000e+00" in data_lines(tmp_path / "fe")       # the max-pop frame


@pytest.mark.parametrize("dtype", ["<f4", "<f8"])
def test_cli_reads_npy_coordinates(tmp_path, oracle, dtype):
    """-f coords.npy (an extension of this build): the same outputs as for the
    float32 matrix itself; float64 files are rounded to float32 first."""
    c64 = gaussian_blobs(2000, 7, seed=44).astype(np.float64) + (1e-9 if dtype == "<f8" else 0.0)
    np.save(tmp_path / "coords.npy", c64.astype(dtype))
    c = c64.astype(dtype).astype(np.float32)
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords.npy"), "-r", "0.12", "-p", str(tmp_path / "pop"),
                        "-b", str(tmp_path / "nn"), "-v"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "2000x7 (.npy)" in r.stdout + r.stderr
    pops = oracle.populations(c, [0.12])[0]
    nn = oracle.nearest_neighbors(c, oracle.free_energies(pops))
    assert data_lines(tmp_path / "pop") == [str(int(p)) for p in pops]
    want = ["%d %s %d %s" % (nn[0][i], fmt_g(nn[1][i]), nn[2][i], fmt_g(nn[3][i])) for i in range(len(c))]
    assert data_lines(tmp_path / "nn") == want
    # a 1-D array is refused with a message, not parsed as text
    np.save(tmp_path / "flat.npy", np.zeros(12, dtype=np.float32))
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "flat.npy"), "-r", "0.1", "-p", str(tmp_path / "p2")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "2-dimensional" in r.stderr + r.stdout


def test_cli_parses_large_ascii_files_in_pieces(tmp_path, oracle):
    """files above 8 MB are parsed by several threads, cut at line ends: same matrix, same outputs; a token
    that is not a number (the reference's `ifs >> float` fails at it and reads garbage from there on,
    tools.hxx:80-108) makes the reader refuse the file, wherever in the pieces it sits"""
    c = write_coords(tmp_path / "coords", gaussian_blobs(260000, 4, seed=47))
    assert os.path.getsize(tmp_path / "coords") > (2 << 22)
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-r", "0.03", "-p", str(tmp_path / "pop"), "-v"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "260000x4" in r.stdout + r.stderr
    assert data_lines(tmp_path / "pop") == [str(int(p)) for p in oracle.populations(c, [0.03])[0]]
    # a bad token two thirds into the file: the frames before it are the data set
    lines = open(tmp_path / "coords").read().split("\n")
    lines[170000] = "0.1 0.2 oops 0.4"
    open(tmp_path / "broken", "w").write("\n".join(lines))
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "broken"), "-r", "0.03", "-p", str(tmp_path / "pop2"), "-v"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "260000 non-empty lines of 4 columns but 680002 readable numbers" in r.stderr


def test_cli_multi_radius_files(tmp_path, oracle):
    c = write_coords(tmp_path / "coords", gaussian_blobs(2000, 10, seed=43))
    radii = [0.3, 0.1, 0.2]
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-R", *[str(x) for x in radii],
                        "-p", str(tmp_path / "pop"), "-d", str(tmp_path / "fe")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    pops = oracle.populations(c, radii)
    for k, rad in enumerate(radii):
        name = "_%f" % np.float32(rad)
        assert data_lines(str(tmp_path / "pop") + name) == [str(int(p)) for p in pops[k]]
        assert data_lines(str(tmp_path / "fe") + name) == [fmt_e(v) for v in oracle.free_energies(pops[k])]


def test_cli_multi_radius_wide_rows_one_sweep(tmp_path):
    """-R with five radii on 52 000 x 30 (.npy in): wide rows with several radii are answered by ONE shared-operand
    sweep (dc_mfma_shared.hpp) inside the command line's session; files per radius as for the narrow case."""
    from oracle.oracle import Oracle
    fast = Oracle()
    c = gaussian_blobs(52000, 30, seed=53)
    np.save(tmp_path / "c.npy", c)
    radii = [0.62, 0.45, 0.55, 0.5, 0.58]
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "c.npy"), "-R", *[str(x) for x in radii],
                        "-p", str(tmp_path / "pop")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    pops = fast.populations(c, radii)
    for k, rad in enumerate(radii):
        assert data_lines(str(tmp_path / "pop") + "_%f" % np.float32(rad)) == [str(int(p)) for p in pops[k]]


def test_cli_without_radius_uses_lumping_radius_and_reuse(tmp_path, oracle):
    """no -r: provisional pop(r=1)+FE+NN pass, radius = sqrt(4 sigma^2) (density_clustering.cpp:649-673);
    then -D/-B re-use (file-level checkpoint/resume)."""
    c = write_coords(tmp_path / "coords", gaussian_blobs(1500, 5, seed=44))
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-p", str(tmp_path / "pop"),
                        "-d", str(tmp_path / "fe"), "-b", str(tmp_path / "nn")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    p1 = oracle.populations(c, [1.0])[0]
    nn1 = oracle.nearest_neighbors(c, oracle.free_energies(p1))
    radius = oracle.lumping_radius(oracle.sigma2(nn1[1]))
    pops = oracle.populations(c, [radius])[0]
    assert data_lines(tmp_path / "pop") == [str(int(p)) for p in pops]
    cm = comments(tmp_path / "pop")
    assert cm["clustering_radius"] == "%.5f" % radius and cm["lumping_radius"] == "%.5f" % radius
    # re-use the free energies (-D) to recompute only the neighbours (-b)
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-D", str(tmp_path / "fe"),
                        "-b", str(tmp_path / "nn2")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    fe_file = np.array([float(x) for x in data_lines(tmp_path / "fe")], dtype=np.float32)
    nn = oracle.nearest_neighbors(c, fe_file)     # FE as printed with 7 digits, like the reference re-reads it
    want = ["%d %s %d %s" % (nn[0][i], fmt_g(nn[1][i]), nn[2][i], fmt_g(nn[3][i])) for i in range(len(c))]
    assert data_lines(tmp_path / "nn2") == want


@pytest.mark.parametrize("full_graph", [False, True])
def test_cli_screening_matches_the_quadratic_restatement(tmp_path, oracle, full_graph):
    """-T FROM STEP TO -o: the GPU's spanning forest of the radius graph (default) or its full pair list
    (DC_SCREENING_FULL_GRAPH=1) + the reference's name bookkeeping must give,
    threshold by threshold, the clustering of the line-by-line restatement (oracle/screening_oracle.cpp:
    O(M^2) scans per threshold, explicit renaming loops), chained through the thresholds like
    density_clustering.cpp:801-812 does."""
    from oracle.oracle import ScreeningOracle
    so = ScreeningOracle()
    c = write_coords(tmp_path / "coords", gaussian_blobs(2500, 3, seed=45))
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-r", "0.05", "-T", "0.5", "0.75", "5.0",
                        "-o", str(tmp_path / "clust"), "-v"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, DC_SCREENING_FULL_GRAPH="1" if full_graph else "0"))
    assert r.returncode == 0, r.stderr + r.stdout
    assert ("within the lumping radius" if full_graph else "span the graph") in r.stdout + r.stderr
    pops = oracle.populations(c, [0.05])[0]
    fe = oracle.free_energies(pops)
    nn = oracle.nearest_neighbors(c, fe)
    clustering = None
    t, n_files = np.float32(0.5), 0
    t_to, step = np.float32(5.0), np.float32(0.75)
    while t < t_to - step / np.float32(10.0) + step and not (t_to + step / np.float32(10.0) + step < t):
        clustering = so.screening(fe, nn[1], t, c, clustering)
        got = data_lines(str(tmp_path / "clust") + ".%0.2f" % t)
        assert got == [str(int(v)) for v in clustering], f"threshold {t}"
        n_files += 1
        t = np.float32(t + step)
    assert n_files == 7
    assert clustering.max() >= 2          # more than one state at the top threshold
    cm = comments(str(tmp_path / "clust") + ".0.50")
    assert cm["screening_from"] == "%.5f" % 0.5 and cm["screening_step"] == "%.5f" % 0.75
    assert cm["screening_to"] == "%.5f" % 5.0


def test_cli_microstates_from_initial_states(tmp_path, oracle):
    """-i initial -o out: frames without a state take the state of their nearest neighbour of lower free
    energy in order of free energy, then states are renamed by population (density_clustering.cpp:345-360,
    458-493)."""
    from oracle.oracle import ScreeningOracle
    so = ScreeningOracle()
    c = write_coords(tmp_path / "coords", gaussian_blobs(2000, 4, seed=46))
    pops = oracle.populations(c, [0.08])[0]
    fe = oracle.free_energies(pops)
    nn = oracle.nearest_neighbors(c, fe)
    initial = so.screening(fe, nn[1], 1.0, c)
    (tmp_path / "initial").write_text("# initial states\n" + "\n".join(str(int(v)) for v in initial) + "\n")
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-r", "0.08", "-i", str(tmp_path / "initial"),
                        "-o", str(tmp_path / "micro")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = so.sorted_names(so.assign_low_density(initial, nn[2], fe))
    assert data_lines(tmp_path / "micro") == [str(int(v)) for v in want]


def test_hosts_say_which_merge_ran(tmp_path, oracle):
    """The reference's merge (density_clustering_cuda.cu:152-180) is never silent about what it does: a multi-device
    session of the C++ hosts that merges through the host instead of RCCL says so on stderr (forced here with two
    "devices" on the one GPU), the command line names the merge under -v in every mode."""
    c = write_coords(tmp_path / "coords", gaussian_blobs(2500, 5, seed=48))
    pops = oracle.populations(c, [0.1])[0]
    env = dict(os.environ, DC_SESSION_DEVICES="0,0", DC_SESSION_ALLOW_DUPLICATE_DEVICES="1")
    cmd = [CLI, "density", "-f", str(tmp_path / "coords"), "-r", "0.1", "-p", str(tmp_path / "pop")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "warning: 2 devices: partial results merge THROUGH THE HOST over PCIe, not RCCL (a device is listed more than once" in r.stderr
    assert data_lines(tmp_path / "pop") == [str(int(p)) for p in pops]       # (two segments, merged on the host)
    r = subprocess.run(cmd + ["-v"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "THROUGH THE HOST" not in r.stderr
    assert "merge: " in r.stdout and ("nothing to merge" in r.stdout or "RCCL" in r.stdout)
    # the shim (Clustering::Density::CUDA::calculate_populations on its resident session)
    c2 = gaussian_blobs(600, 10, seed=49)
    fe = oracle.free_energies(oracle.populations(c2, [0.2])[0])
    c2.tofile(tmp_path / "c.f32")
    fe.tofile(tmp_path / "fe.f32")
    r = subprocess.run([SHIM, str(tmp_path / "c.f32"), "600", "10", str(tmp_path / "fe.f32"), "0.2"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr
    assert "warning: 2 devices: partial results merge THROUGH THE HOST" in r.stderr


def test_cli_output_needs_a_mode(tmp_path):
    (tmp_path / "coords").write_text("0 0\n1 1\n0.5 0.5\n")
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "coords"), "-r", "1", "-o", str(tmp_path / "x")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "one of -T/-i is needed" in r.stderr


def test_cpp_shim_reference_signatures(tmp_path, oracle):
    c = gaussian_blobs(1200, 10, seed=45)
    n, d = c.shape
    radii = [0.2, 0.3]
    pops = oracle.populations(c, radii)
    fe = oracle.free_energies(pops[0])
    c.tofile(tmp_path / "c.f32")
    fe.tofile(tmp_path / "fe.f32")
    r = subprocess.run([SHIM, str(tmp_path / "c.f32"), str(n), str(d), str(tmp_path / "fe.f32"), "0.3", "0.2"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.splitlines()
    assert lines[0].startswith("gpus ")
    got = {l.split()[1]: [int(x) for x in l.split()[2:]] for l in lines if l.startswith("pops ")}
    assert got["%.9g" % np.float32(0.2)] == pops[0].tolist()
    assert got["%.9g" % np.float32(0.3)] == pops[1].tolist()
    part = {l.split()[1]: [int(x) for x in l.split()[2:]] for l in lines if l.startswith("part ")}
    want = oracle.populations(c, radii, n // 3, n // 2)
    assert part["%.9g" % np.float32(0.2)] == want[0].tolist()
    exp = oracle.nearest_neighbors(c, fe)
    for l in lines:
        if l.startswith("nn "):
            _, i, a, b, cc, dd = l.split()
            i = int(i)
            assert int(a) == exp[0][i] and int(cc) == exp[2][i]
            assert np.float32(float(b)) == exp[1][i] and np.float32(float(dd)) == exp[3][i]
    # CUDA::screening of the shim, three chained thresholds, against the quadratic restatement
    from oracle.oracle import ScreeningOracle
    so = ScreeningOracle()
    want = None
    screens = [l.split() for l in lines if l.startswith("screen ")]
    assert [s[1] for s in screens] == ["0.5", "1.5", "3"]
    for sline in screens:
        want = so.screening(fe, exp[1], float(sline[1]), c, want)
        assert [int(x) for x in sline[2:]] == want.tolist()
    # ... and from an initial clustering that no lower threshold produced (full radius graph)
    foreign = np.where(fe < np.float32(1.0), 1 + np.arange(len(fe)) % 3, 0)
    fline = [l.split() for l in lines if l.startswith("foreign ")][0]
    assert [int(x) for x in fline[2:]] == so.screening(fe, exp[1], 2.0, c, foreign).tolist()


def test_shim_neighbourhood_cache_sees_in_place_edits(tmp_path, oracle):
    """CUDA::high_density_neighborhood (declared density_clustering_cuda.hpp:56-62, CPU semantics density_clustering.cpp:
    292-332) serves every frame of a screening pass from one cached radius graph.  Between passes the buffers may be
    rewritten IN PLACE -- one row of the coordinates, two entries of the order -- and the next pass must see it: every
    pass against the brute-force neighbourhoods of the arrays as they are then (the sampled key missed it)."""
    from refmath import d2_matrix
    c = gaussian_blobs(400, 10, seed=51)
    n, d = c.shape
    fe = oracle.free_energies(oracle.populations(c, [0.2])[0])
    c.tofile(tmp_path / "c.f32")
    fe.tofile(tmp_path / "fe.f32")
    max_dist = np.float32(0.06)
    r = subprocess.run([SHIM, str(tmp_path / "c.f32"), str(n), str(d), str(tmp_path / "fe.f32"), "hdn", "%.9g" % max_dist],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = [l.split() for l in r.stdout.splitlines()]
    orders = {l[0]: np.array([int(x) for x in l[1:]]) for l in lines if l[0].startswith("order_")}
    order = orders["order_hdn1"]                              # (std::sort: ties of free energy fall where it puts them)
    assert sorted(order.tolist()) == list(range(n)) and (np.diff(fe[order]) >= 0).all()
    moved = int([l for l in lines if l[0] == "moved"][0][1])
    assert moved == order[3] and (orders["order_hdn2"] == order).all()

    def expect(coords, order):
        d2 = d2_matrix(coords[order])                          # [position, position]
        return [sorted(set(np.nonzero(d2[i] < max_dist)[0].tolist()) | {i}) for i in range(n)]

    def got(tag):
        return [[int(x) for x in l[2:]] for l in lines if l[0] == tag]

    assert got("hdn1") == expect(c, order)
    c2 = c.copy()
    c2[moved] += np.float32(100.0)
    want2 = expect(c2, order)
    assert want2 != expect(c, order) and got("hdn2") == want2
    order3 = order.copy()
    order3[[5, 9]] = order3[[9, 5]]
    assert (orders["order_hdn3"] == order3).all()
    assert got("hdn3") == expect(c2, order3) and got("hdn3") != want2


def test_cli_refuses_malformed_coordinate_files(tmp_path):
    """see tests/test_cli_cpu.py::test_malformed_coordinate_files_are_refused -- here with a GPU present, so
    that the reader itself is reached"""
    cases = {"ragged": "0 1\n2 3 4\n5 6\n", "comment": "# x y\n0 1\n2 3\n", "nan": "0 1\nnan 3\n",
             "inf": "0 1\n2 inf\n", "hex": "0 1\n0x1p3 3\n", "short": "0 1\n2\n"}
    for name, text in cases.items():
        (tmp_path / name).write_text(text)
        r = subprocess.run([CLI, "density", "-f", str(tmp_path / name), "-r", "1", "-p", str(tmp_path / "pop")],
                           capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and "readable numbers" in r.stderr, (name, r.stderr)
    (tmp_path / "ok").write_text("0 1\n\n2 3\n   4   5\n")     # empty lines are skipped, blanks are separators
    r = subprocess.run([CLI, "density", "-f", str(tmp_path / "ok"), "-r", "3", "-p", str(tmp_path / "pop")],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert data_lines(tmp_path / "pop") == ["2", "3", "2"]     # d2 = 8, 8, 32 against r2 = 9
