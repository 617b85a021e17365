// clustering_cli.cpp -- `clustering density` on MI355X.
//
// Re-creates the density mode of the reference's command line (clustering.cpp:141-193 options,
// density_clustering.cpp:559-825 control flow) on top of libdcdensity.so, with the reference's
// file formats:
//   coords   whitespace separated ASCII, one frame per line              (tools.hxx:39-111)
//   -p FILE  header + one integer population per line                    (tools.cpp:50-56)
//   -d FILE  header + one free energy per line, "%e"                     (tools.cpp:42-48)
//   -b FILE  header + "id(nn) dsqr(nn) id(nn_hd) dsqr(nn_hd)" per line   (tools.cpp:144-174)
//   -R r1 r2 ...  several radii in one sweep, files <base>_%f            (density_clustering.cpp:633-642)
//   "#@   key = %.5f" header lines carry clustering_radius / lumping_radius between stages
//   (tools.cpp:229-277) and are honoured when -D / -B re-use earlier results.
// Boost is not available here, so the options are parsed by hand; the option names, their
// meaning and the error texts follow the reference.  Screening (-T), clustering output (-o) and
// microstates from initial states (-i) run on the GPU-built radius graph (screening_host.cpp).
// There is no CPU implementation behind this binary: without a HIP device it exits like the
// reference's CUDA build does (clustering.cpp:110-113).
#include "../../include/dc_density.h"
#include "density_clustering_hip.hpp"
#include "screening_host.hpp"

#include <cfloat>
#include <algorithm>
#include <thread>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iostream>
#include <limits>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace {

const char* kVersion = "amd-hip 0.1 (moldyn/Clustering v1.3 file formats)";
bool g_verbose = false;

#define LOG(...)                        \
  do {                                  \
    if (g_verbose) {                    \
      std::printf(__VA_ARGS__);         \
      std::fflush(stdout);              \
    }                                   \
  } while (0)

[[noreturn]] void die(const std::string& msg) {
  std::cerr << msg << std::endl;
  std::exit(EXIT_FAILURE);
}

const char* kHelp =
    "clustering density: \n"
    "perform clustering of MD data based on phase space densities.\n"
    "densities are approximated by counting neighboring frames inside\n"
    "a n-dimensional hypersphere of specified radius.\n"
    "distances are measured with n-dim P2-norm.\n\n"
    "options:\n"
    "  -h [ --help ]                         show this help.\n"
    "  -f [ --file ] arg                     input (required): phase space coordinates\n"
    "                                        (space separated ASCII; this build also\n"
    "                                        reads NumPy .npy, float32/float64 C order).\n"
    "  -r [ --radius ] arg                   parameter: hypersphere radius. If not used, the\n"
    "                                        lumping radius will be used instead.\n"
    "  -R [ --radii ] arg                    parameter: list of radii for population/free energy\n"
    "                                        calculations (several radii in one go).\n"
    "  -p [ --population ] arg               output (optional): population per frame (if -R is set:\n"
    "                                        this defines only the basename).\n"
    "  -d [ --free-energy ] arg              output (optional): free energies per frame\n"
    "                                        (if -R is set: this defines only the basename).\n"
    "  -D [ --free-energy-input ] arg        input (optional): reuse free energy info.\n"
    "  -b [ --nearest-neighbors ] arg        output (optional): nearest neighbor info.\n"
    "  -B [ --nearest-neighbors-input ] arg  input (optional): reuse nearest neighbor info.\n"
    "  -T [ --threshold-screening ] arg      parameters: screening of free energy landscape. format:\n"
    "                                        FROM STEP TO; e.g.: \'-T 0.1 0.1 11.1\'. for defaults:\n"
    "                                        \'-T -1\'. FROM: 0.1, STEP: 0.1, TO: MAX_FE.\n"
    "  -o [ --output ] arg                   output (optional): clustering information.\n"
    "  -i [ --input ] arg                    input (optional): initial state definition.\n"
    "  -n [ --nthreads ] arg (=0)            accepted for compatibility (the sweeps run on the GPU).\n"
    "  -v [ --verbose ]                      verbose mode: print runtime information to STDOUT.\n";

struct Options {
  std::string file, population, free_energy, free_energy_input, nn, nn_input;
  bool has_radius = false;
  float radius = 0.0f;
  std::vector<float> radii;
  std::string output, input;
  bool has_threshold = false;
  std::vector<float> threshold;
};

bool is_number(const char* s) {
  char* end = nullptr;
  std::strtof(s, &end);
  return end != s && *end == '\0';
}

Options parse(int argc, char** argv) {
  Options o;
  std::map<std::string, std::string> longnames = {
      {"--file", "-f"}, {"--radius", "-r"}, {"--radii", "-R"}, {"--population", "-p"},
      {"--free-energy", "-d"}, {"--free-energy-input", "-D"}, {"--nearest-neighbors", "-b"},
      {"--nearest-neighbors-input", "-B"}, {"--nthreads", "-n"}, {"--verbose", "-v"},
      {"--help", "-h"}, {"--threshold-screening", "-T"}, {"--output", "-o"}, {"--input", "-i"}};
  for (int i = 2; i < argc; ++i) {
    std::string a = argv[i];
    if (longnames.count(a)) a = longnames[a];
    auto need = [&](const char* what) -> const char* {
      if (i + 1 >= argc) die(std::string("\nerror parsing arguments:\n\nthe required argument for option '") + what + "' is missing\n");
      return argv[++i];
    };
    if (a == "-h") {
      std::cout << kHelp << std::endl;
      std::exit(EXIT_SUCCESS);
    } else if (a == "-f") o.file = need("--file");
    else if (a == "-r") {
      o.radius = std::strtof(need("--radius"), nullptr);
      o.has_radius = true;
    } else if (a == "-R") {
      while (i + 1 < argc && is_number(argv[i + 1])) o.radii.push_back(std::strtof(argv[++i], nullptr));
      if (o.radii.empty()) die("\nerror parsing arguments:\n\nthe required argument for option '--radii' is missing\n");
    } else if (a == "-p") o.population = need("--population");
    else if (a == "-d") o.free_energy = need("--free-energy");
    else if (a == "-D") o.free_energy_input = need("--free-energy-input");
    else if (a == "-b") o.nn = need("--nearest-neighbors");
    else if (a == "-B") o.nn_input = need("--nearest-neighbors-input");
    else if (a == "-n") (void)need("--nthreads");
    else if (a == "-v") g_verbose = true;
    else if (a == "-T") {
      o.has_threshold = true;
      while (i + 1 < argc && is_number(argv[i + 1])) o.threshold.push_back(std::strtof(argv[++i], nullptr));
      if (o.threshold.empty()) die("\nerror parsing arguments:\n\nthe required argument for option '--threshold-screening' is missing\n");
    } else if (a == "-o") o.output = need("--output");
    else if (a == "-i") o.input = need("--input");
    else {
      die("\nerror parsing arguments:\n\nunrecognised option '" + a + "'\n\n" + kHelp);
    }
  }
  if (o.file.empty()) die(std::string("\nerror parsing arguments:\n\nthe option '--file' is required but missing\n\n") + kHelp);
  return o;
}

// ---- IO -------------------------------------------------------------------------------------

// NumPy .npy (format 1.0 - 3.0): a C-ordered 2-D array of little-endian float32 or float64 (cast to
// float like the reference's `ifs >> float` would round a decimal).  An extension of this build
// (SURVEY.md section 8(f) rank 3): the reference reads ASCII only (tools.hxx:39-111, two passes).
bool read_npy(const std::string& text, const std::string& fname, std::vector<float>& coords,
              std::size_t& n_rows, std::size_t& n_cols) {
  static const char magic[6] = {'\x93', 'N', 'U', 'M', 'P', 'Y'};
  if (text.size() < 10 || std::memcmp(text.data(), magic, 6) != 0) return false;
  const unsigned major = (unsigned char)text[6];
  std::size_t hlen = 0, hoff = 0;
  if (major == 1) {
    hlen = (unsigned char)text[8] | ((std::size_t)(unsigned char)text[9] << 8);
    hoff = 10;
  } else {
    if (text.size() < 12) die("error: truncated .npy header in '" + fname + "'");
    for (int b = 0; b < 4; ++b) hlen |= (std::size_t)(unsigned char)text[8 + b] << (8 * b);
    hoff = 12;
  }
  if (text.size() < hoff + hlen) die("error: truncated .npy header in '" + fname + "'");
  const std::string hdr = text.substr(hoff, hlen);
  auto field = [&](const std::string& key) {
    const std::size_t k = hdr.find("'" + key + "'");
    if (k == std::string::npos) die("error: .npy header of '" + fname + "' lacks '" + key + "'");
    return hdr.find(':', k) + 1;
  };
  std::size_t d = field("descr");
  const std::size_t q0 = hdr.find('\'', d), q1 = hdr.find('\'', q0 + 1);
  const std::string descr = hdr.substr(q0 + 1, q1 - q0 - 1);
  if (descr != "<f4" && descr != "<f8" && descr != "|f4")
    die("error: '" + fname + "': .npy dtype '" + descr + "' not supported (need <f4 or <f8)");
  if (hdr.substr(field("fortran_order")).find("False") > 2)
    die("error: '" + fname + "': Fortran-ordered .npy arrays are not supported");
  const std::size_t sh = hdr.find('(', field("shape"));
  char* e1 = nullptr;
  const unsigned long long r = std::strtoull(hdr.c_str() + sh + 1, &e1, 10);
  while (*e1 == ',' || *e1 == ' ') ++e1;
  char* e2 = nullptr;
  const unsigned long long c = std::strtoull(e1, &e2, 10);
  if (e2 == e1) die("error: '" + fname + "': .npy array must be 2-dimensional [frames][columns]");
  while (*e2 == ',' || *e2 == ' ') ++e2;
  if (*e2 != ')') die("error: '" + fname + "': .npy array must be 2-dimensional [frames][columns]");
  const std::size_t item = (descr == "<f8") ? 8 : 4;
  n_rows = (std::size_t)r;
  n_cols = (std::size_t)c;
  if (text.size() - hoff - hlen < n_rows * n_cols * item) die("error: truncated .npy data in '" + fname + "'");
  coords.resize(n_rows * n_cols);
  const char* data = text.data() + hoff + hlen;
  if (item == 4) {
    std::memcpy(coords.data(), data, coords.size() * 4);
  } else {
    for (std::size_t i = 0; i < coords.size(); ++i) {
      double v;
      std::memcpy(&v, data + 8 * i, 8);
      coords[i] = (float)v;
    }
  }
  return true;
}

// whitespace separated ASCII matrix; n_cols from the first non-empty line (tools.hxx:52-76) -- or a
// .npy file (recognised by its magic bytes)
void read_coords(const std::string& fname, std::vector<float>& coords, std::size_t& n_rows,
                 std::size_t& n_cols) {
  std::ifstream ifs(fname, std::ios::binary);
  if (ifs.fail()) die("error: cannot open file '" + fname + "'");
  LOG("~~~ reading coordinates\n    from file: %s\n", fname.c_str());
  // (one block read: the character-wise istreambuf copy took 0.3 s of the 130 MB of C3)
  ifs.seekg(0, std::ios::end);
  const std::streamoff file_size = ifs.tellg();
  ifs.seekg(0, std::ios::beg);
  std::string text;
  if (file_size > 0) {
    text.resize((std::size_t)file_size);
    ifs.read(&text[0], file_size);
    text.resize((std::size_t)ifs.gcount());
  } else {   // (not seekable: a pipe)
    text.assign((std::istreambuf_iterator<char>(ifs)), std::istreambuf_iterator<char>());
  }
  n_rows = 0;
  n_cols = 0;
  if (read_npy(text, fname, coords, n_rows, n_cols)) {
    LOG("    with dimensions: %zux%zu (.npy)\n\n", n_rows, n_cols);
    return;
  }
  const char* p = text.c_str();
  const char* end = p + text.size();
  // first non-empty line fixes n_cols
  {
    const char* q = p;
    while (q < end) {
      const char* eol = static_cast<const char*>(std::memchr(q, '\n', end - q));
      if (!eol) eol = end;
      std::string line(q, eol);
      std::istringstream ss(line);
      std::string tok;
      std::size_t n = 0;
      while (ss >> tok) ++n;
      if (n > 0) {
        n_cols = n;
        break;
      }
      q = eol + 1;
    }
  }
  if (n_cols == 0) die("error: opened empty file '" + fname + "'");
  // The text is cut at line ends into one piece per thread (the reference's `ifs >> float` loop,
  // tools.hxx:80-108, is a single pass; 130 MB of C3 take 0.7 s that way).  A piece stops at the first
  // token that is not a number -- like `ifs >> buf` -- and everything after that piece is dropped.
  auto is_space = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '\v' || ch == '\f'; };
  const std::size_t n_pieces = std::max<std::size_t>(
      1, std::min<std::size_t>({(std::size_t)std::thread::hardware_concurrency(), 16, text.size() >> 22}));
  std::vector<const char*> cut(n_pieces + 1, end);
  cut[0] = p;
  for (std::size_t i = 1; i < n_pieces; ++i) {
    const char* q = p + i * (text.size() / n_pieces);
    q = static_cast<const char*>(std::memchr(q, '\n', end - q));
    cut[i] = q ? q + 1 : end;
  }
  std::vector<std::vector<float>> piece(n_pieces);
  std::vector<char> stopped(n_pieces, 0);
  auto parse = [&](std::size_t i) {
    const char* q = cut[i];
    const char* stop = cut[i + 1];
    std::vector<float>& out = piece[i];
    out.reserve((std::size_t)(stop - q) / 8);
    while (q < stop) {
      if (is_space(*q)) {   // (strtof would skip white space across the cut and take the next piece's number)
        ++q;
        continue;
      }
      char* next = nullptr;
      const float v = std::strtof(q, &next);   // (this program never calls setlocale: "C" locale)
      // what `ifs >> float` takes is a plain decimal number: strtof's "nan", "inf" and hex floats are
      // not numbers to the reference (tools.hxx:99-101) and stop the read here as well
      bool plain = next != q;
      for (const char* r = q; plain && r < next; ++r)
        plain = (*r >= '0' && *r <= '9') || *r == '+' || *r == '-' || *r == '.' || *r == 'e' || *r == 'E';
      if (!plain) {
        stopped[i] = 1;
        break;
      }
      out.push_back(v);
      q = next;
    }
  };
  if (n_pieces == 1) {
    parse(0);
  } else {
    std::vector<std::thread> th;
    for (std::size_t i = 0; i < n_pieces; ++i) th.emplace_back(parse, i);
    for (auto& t : th) t.join();
  }
  coords.clear();
  std::size_t total = 0;
  for (std::size_t i = 0; i < n_pieces; ++i) {
    total += piece[i].size();
    if (stopped[i]) break;
  }
  coords.reserve(total);
  for (std::size_t i = 0; i < n_pieces; ++i) {
    coords.insert(coords.end(), piece[i].begin(), piece[i].end());
    if (stopped[i]) break;
  }
  // rows = non-empty lines, like the reference counts them (tools.hxx:66-72); it then reads rows x
  // columns numbers in stream order whatever the line structure is, and garbage once the stream fails.
  // A file whose numbers do not fill that shape exactly is refused here instead.
  std::size_t n_lines = 0;
  for (const char* q = p; q < end;) {
    const char* eol = static_cast<const char*>(std::memchr(q, '\n', end - q));
    if (!eol) eol = end;
    n_lines += (eol > q) ? 1 : 0;
    q = eol + 1;
  }
  n_rows = n_lines;
  if (coords.size() != n_rows * n_cols) {
    char msg[256];
    std::snprintf(msg, sizeof(msg), "error: '%s' has %zu non-empty lines of %zu columns but %zu readable numbers",
                  fname.c_str(), n_rows, n_cols, coords.size());
    die(msg);
  }
  LOG("    with dimensions: %zux%zu\n\n", n_rows, n_cols);
}

std::string provenance_header(int argc, char** argv) {
  std::ostringstream h;
  time_t raw;
  time(&raw);
  h << "# clustering " << kVersion << " - " << argv[1] << "\n#\n# Created " << asctime(localtime(&raw))
    << "# by following command:\n#\n# ";
  for (int i = 0; i < argc; ++i) h << argv[i] << " ";
  h << "\n#\n# MI355X-native implementation of the density hot path of moldyn/Clustering\n"
    << "# (method: Sittel & Stock, J. Chem. Theory Comput. 12, 2426 (2016); https://github.com/moldyn/clustering)\n";
  return h.str();
}

typedef std::map<std::string, float> Comments;

// "#@   key = %.5f" lines, only non-zero values (tools.cpp:267-277)
std::string with_comments(std::string header, const Comments& cm) {
  header.append("#\n# The following comments are reused for identifying\n# user-based mistakes and should not be modified.\n");
  for (const auto& kv : cm)
    if (kv.second != 0.) {
      char buf[256];
      std::snprintf(buf, sizeof(buf), "#@   %s = %.5f\n", kv.first.c_str(), kv.second);
      header.append(buf);
    }
  return header;
}

void read_comments(const std::string& fname, Comments& cm) {
  std::ifstream ifs(fname);
  if (ifs.fail()) die("error: cannot open file '" + fname + "'");
  std::string line;
  while (std::getline(ifs, line)) {
    if (line.compare(0, 2, "#@") != 0) continue;
    std::istringstream ss(line.substr(2));
    std::string key, eq;
    float val;
    if (!(ss >> key >> eq >> val) || eq != "=") continue;
    auto it = cm.find(key);
    if (it == cm.end()) continue;
    if (it->second != 0 && std::abs(it->second - val) > 0.001)
      LOG("warning: the values of %s are not in agreement\n        %g vs. %g\n", key.c_str(), val, it->second);
    it->second = val;
  }
}

FILE* open_out(const std::string& fname) {
  FILE* f = std::fopen(fname.c_str(), "w");
  if (!f) die("error: cannot open file '" + fname + "' for writing.");
  return f;
}

void write_pops(const std::string& fname, const std::uint32_t* pops, std::size_t n,
                const std::string& header, const Comments& cm) {
  FILE* f = open_out(fname);
  std::fputs((with_comments(header, cm) + "#\n# point density of each frame\n").c_str(), f);
  for (std::size_t i = 0; i < n; ++i) std::fprintf(f, "%u\n", pops[i]);
  std::fclose(f);
}

void write_fes(const std::string& fname, const float* fe, std::size_t n, const std::string& header,
               const Comments& cm) {
  FILE* f = open_out(fname);
  std::fputs((with_comments(header, cm) + "#\n# free energy of each frame\n").c_str(), f);
  for (std::size_t i = 0; i < n; ++i) std::fprintf(f, "%e\n", fe[i]);   // std::scientific, 6 digits
  std::fclose(f);
}

// tools.cpp:63-69
void write_clustered_trajectory(const std::string& fname, const std::vector<std::size_t>& traj,
                                const std::string& header, const Comments& cm) {
  FILE* f = open_out(fname);
  std::fputs((with_comments(header, cm) + "#\n# state/cluster id frames are assigned to\n").c_str(), f);
  for (std::size_t v : traj) std::fprintf(f, "%zu\n", v);
  std::fclose(f);
}

void write_neighborhood(const std::string& fname, std::size_t n, const std::uint32_t* nn_idx,
                        const float* nn_d2, const std::uint32_t* hd_idx, const float* hd_d2,
                        const std::string& header, const Comments& cm) {
  FILE* f = open_out(fname);
  std::fputs((with_comments(header, cm) +
              "#\n# column definitions:\n"
              "#        nn = nearest neighbor\n"
              "#     nn_hd = nearest neighbor with higher density\n"
              "#     id(i) = id/line number of i\n"
              "#   dsqr(i) = squared euclidean distance to i\n#\n"
              "# id(nn)  dsqr(nn) id(nn_hd) dsqr(nn_hd)\n").c_str(), f);
  for (std::size_t i = 0; i < n; ++i)   // default ostream float format == %g with 6 digits
    std::fprintf(f, "%u %g %u %g\n", nn_idx[i], nn_d2[i], hd_idx[i], hd_d2[i]);
  std::fclose(f);
}

// one number per line, lines that do not start with a number are comments (tools.hxx:230-252)
template <typename T>
std::vector<T> read_single_column(const std::string& fname) {
  std::ifstream ifs(fname);
  if (ifs.fail()) die("error: cannot open file '" + fname + "'");
  std::vector<T> dat;
  std::string line;
  while (std::getline(ifs, line)) {
    std::istringstream ss(line);
    T v;
    if (ss >> v) dat.push_back(v);
  }
  if (dat.empty()) die("error: opened empty file '" + fname + "'");
  return dat;
}

void read_neighborhood(const std::string& fname, std::vector<std::uint32_t>& nn_idx,
                       std::vector<float>& nn_d2, std::vector<std::uint32_t>& hd_idx,
                       std::vector<float>& hd_d2) {
  std::ifstream ifs(fname);
  if (ifs.fail()) die("error: cannot open file '" + fname + "'");
  std::string line;
  while (std::getline(ifs, line)) {
    std::istringstream ss(line);
    std::size_t a, c;
    float b, d;
    if (ss >> a >> b >> c >> d) {
      nn_idx.push_back((std::uint32_t)a);
      nn_d2.push_back(b);
      hd_idx.push_back((std::uint32_t)c);
      hd_d2.push_back(d);
    }
  }
}

std::string sprintf_f(const std::string& base, float v) {
  char buf[64];
  std::snprintf(buf, sizeof(buf), "_%f", v);
  return base + buf;
}

// ---- the density mode (density_clustering.cpp:559-825 without screening) ------------------------
void must(int rc, const char* what) {
  if (rc != DC_OK) die(std::string("HIP error: ") + what + "\n" + dc_hip_last_error());
}

int density_main(int argc, char** argv) {
  Options o = parse(argc, argv);
  // like the reference's CUDA build: fail early if there is no GPU (clustering.cpp:110-113)
  const int n_gpus = Clustering::Density::CUDA::get_num_gpus();
  const std::string header = provenance_header(argc, argv);
  Comments cm = {{"clustering_radius", 0.f}, {"lumping_radius", 0.f}, {"screening_from", 0.f},
                 {"screening_to", 0.f}, {"screening_step", 0.f}, {"minimal_population", 0.f},
                 {"cmin", 0.f}, {"single_coring_time", 0.f}, {"limits", 0.f}};
  LOG("\n%s\n~~~ using for parallization: HIP (%d GPU%s)\n", header.c_str(), n_gpus, n_gpus == 1 ? "" : "s");

  std::vector<float> coords;
  std::size_t n_rows = 0, n_cols = 0;
  read_coords(o.file, coords, n_rows, n_cols);
  if (n_rows == 0) die("error: no frames in '" + o.file + "'");

  // ONE session for the whole run (density_clustering.cpp:597-817): the coordinates go to the GPUs once
  // and stay there across pop -> FE -> NN -> sigma2 -> (second pop + NN at the lumping radius) -> forest
  dc_hip_session* session = nullptr;
  auto sess = [&]() {
    if (!session) {
      must(dc_hip_session_open(coords.data(), n_rows, n_cols, nullptr, n_gpus, &session), "uploading the coordinates");
      // which merge runs (density_clustering_cuda.cu:152-180 is never silent about its own): a multi-GPU run that fell
      // back from RCCL to the host merge is correct and slow -- always said on stderr, every mode under -v
      if (dc_hip_session_merge_mode(session) == 2) std::cerr << "warning: " << dc_hip_session_merge_note(session) << std::endl;
      LOG("merge: %s\n", dc_hip_session_merge_note(session));
    }
    return session;
  };
  std::vector<float> fe;
  bool fe_resident = false;          // the session holds exactly `fe`
  std::vector<std::uint32_t> pops;   // single-radius populations
  auto sweep = [&](const std::vector<float>& radii, std::size_t fe_index, bool want_nn,
                   std::vector<std::uint32_t>& pops_out, std::vector<float>& fe_out,
                   std::vector<std::uint32_t>& nn_idx, std::vector<float>& nn_d2,
                   std::vector<std::uint32_t>& hd_idx, std::vector<float>& hd_d2) {
    pops_out.assign(radii.size() * n_rows, 0);
    fe_out.assign(n_rows, 0.f);
    must(dc_hip_session_populations(sess(), radii.data(), radii.size(), pops_out.data()), "population sweep");
    must(dc_hip_session_free_energies(sess(), fe_index, fe_out.data(), nullptr), "free energies");
    fe_resident = (&fe_out == &fe);
    if (want_nn) {
      nn_idx.assign(n_rows, 0);
      hd_idx.assign(n_rows, 0);
      nn_d2.assign(n_rows, 0.f);
      hd_d2.assign(n_rows, 0.f);
      must(dc_hip_session_nearest_neighbors(sess(), nn_idx.data(), nn_d2.data(), hd_idx.data(), hd_d2.data(), nullptr),
           "nearest-neighbour sweep");
    }
  };
  auto sigma2_of = [&](const std::vector<float>& nn_d2) {
    double s = 0.0;   // frame order, double (density_clustering.cpp:334-343)
    for (float v : nn_d2) s += (double)v;
    return s / (double)nn_d2.size();
  };
  std::vector<std::uint32_t> nn_idx, hd_idx;
  std::vector<float> nn_d2, hd_d2;
  bool have_nn = false;

  LOG("~~~ free energy and population\n");
  if (!o.free_energy_input.empty()) {
    LOG("    re-using free energy: %s\n", o.free_energy_input.c_str());
    if (!o.radii.empty() || o.has_radius) LOG("warning: radius (-r/-R) is ignored\n");
    if (!o.free_energy.empty() || !o.population.empty()) LOG("warning: -p/-d flags are ignored\n");
    fe = read_single_column<float>(o.free_energy_input);
    if (fe.size() != n_rows) die("error: free energy file does not match the number of frames");
    read_comments(o.free_energy_input, cm);
  } else if (!o.free_energy.empty() || !o.population.empty() || !o.output.empty()) {
    if (!o.radii.empty()) {
      if (!o.output.empty()) die("error: clustering cannot be done with several radii (-R is set).");
      if (o.free_energy.empty() && o.population.empty())
        die("error: no output defined for populations or free energies.\n       why did you define -R ?");
      LOG("    calculating free energy and population\n    using radii: ");
      for (float r : o.radii) LOG("%g, ", r);
      LOG("\n    using HIP\n");
      // all radii in one sweep; free energies per radius from the host formula.  (Ascending: the files are per radius,
      // and the multi-radius sweep leaves out the small radii a chain holds nothing of -- dc_mfma_msym.hpp mr_chain_k.)
      std::vector<float> radii = o.radii;
      std::sort(radii.begin(), radii.end());
      std::vector<std::uint32_t> all;
      std::vector<float> fe_r;
      for (std::size_t k = 0; k < radii.size(); ++k) {
        if (k == 0) {
          sweep(radii, 0, false, all, fe_r, nn_idx, nn_d2, hd_idx, hd_d2);
        } else if (!o.free_energy.empty()) {
          // FE of radius k: same formula on that population row (density_clustering.cpp:197-212)
          const std::uint32_t* p = all.data() + k * n_rows;
          std::uint32_t mx = 0;
          for (std::size_t i = 0; i < n_rows; ++i) mx = p[i] > mx ? p[i] : mx;
          const float rec = 1.0f / (float)mx;
          for (std::size_t i = 0; i < n_rows; ++i) {
            const float q = (float)p[i] * rec;
            fe_r[i] = (float)(-std::log((double)q));
          }
        }
        LOG("    storing results for radius %g\n", radii[k]);
        if (!o.population.empty())
          write_pops(sprintf_f(o.population, radii[k]), all.data() + k * n_rows, n_rows, header, cm);
        if (!o.free_energy.empty())
          write_fes(sprintf_f(o.free_energy, radii[k]), fe_r.data(), n_rows, header, cm);
      }
    } else {
      float radius_lump = 1.0f;
      if (!o.has_radius) {
        // no radius given: provisional pop(r=1)+FE+NN pass to get the lumping radius sqrt(4 sigma^2)
        // (density_clustering.cpp:649-673)
        LOG("    computing lumping radius\n");
        std::vector<std::uint32_t> p1;
        std::vector<float> fe1;
        sweep({radius_lump}, 0, true, p1, fe1, nn_idx, nn_d2, hd_idx, hd_d2);
        radius_lump = (float)std::sqrt(4 * sigma2_of(nn_d2));
        LOG("        d_lump=%g\n", radius_lump);
        cm["lumping_radius"] = radius_lump;
      }
      const float radius = o.has_radius ? o.radius : radius_lump;
      LOG("    calculating free energy and population\n    using radius: %g\n", radius);
      cm["clustering_radius"] = radius;
      const bool want_nn = (!o.nn.empty() || !o.output.empty()) && o.nn_input.empty();
      sweep({radius}, 0, want_nn, pops, fe, nn_idx, nn_d2, hd_idx, hd_d2);
      have_nn = want_nn;
      if (!o.population.empty()) {
        LOG("    storing population in: %s\n", o.population.c_str());
        write_pops(o.population, pops.data(), n_rows, header, cm);
      }
      if (!o.free_energy.empty()) {
        LOG("    storing free energy in: %s\n", o.free_energy.c_str());
        write_fes(o.free_energy, fe.data(), n_rows, header, cm);
      }
    }
  }

  LOG("\n~~~ nearest neighbors\n");
  if (!o.nn_input.empty()) {
    LOG("    re-using nearest neighbor: %s\n", o.nn_input.c_str());
    read_neighborhood(o.nn_input, nn_idx, nn_d2, hd_idx, hd_d2);
    read_comments(o.nn_input, cm);
  } else if (!o.nn.empty() || !o.output.empty()) {
    if (!o.radii.empty())
      die("error: nearest neighbor calculation cannot be done with\n       several radii (-R is set).");
    if (fe.empty())
      die("error: nearest neighbors need free energies: give -p/-d (with -r) or -D.");
    LOG("    calculating nearest neighbors\n");
    if (!have_nn) {
      nn_idx.assign(n_rows, 0);
      hd_idx.assign(n_rows, 0);
      nn_d2.assign(n_rows, 0.f);
      hd_d2.assign(n_rows, 0.f);
      // one segment per GPU, concurrently, merged on the devices (density_clustering_cuda.cu:293-326 shards
      // row blocks); the free energies are resident unless they came from a file (-D)
      if (!fe_resident) must(dc_hip_session_set_free_energies(sess(), fe.data()), "uploading the free energies");
      fe_resident = true;
      must(dc_hip_session_nearest_neighbors(sess(), nn_idx.data(), nn_d2.data(), hd_idx.data(), hd_d2.data(), nullptr),
           "nearest-neighbour sweep");
    }
    if (cm["lumping_radius"] == 0.) {
      const float radius_lump = (float)std::sqrt(4 * sigma2_of(nn_d2));
      LOG("    lumping radius: %g\n", radius_lump);
      cm["lumping_radius"] = radius_lump;
    }
    if (!o.nn.empty()) {
      LOG("    storing nearest neighbors in: %s\n", o.nn.c_str());
      write_neighborhood(o.nn, n_rows, nn_idx.data(), nn_d2.data(), hd_idx.data(), hd_d2.data(), header, cm);
    }
  }

  //// clustering (density_clustering.cpp:739-820)
  if (!o.output.empty()) {
    namespace H = Clustering::Density::HIP;
    if (!o.radii.empty())
      die("error: output needs to depend on single radius\n       but several radii (-R) are set.");
    if (fe.size() != n_rows || nn_d2.size() != n_rows)
      die("error: clustering output needs free energies and nearest neighbors (-r/-D, -B).");
    std::vector<std::size_t> clustering;
    if (!o.input.empty()) {
      LOG("~~~ generating microstates\n");
      if (o.has_threshold) LOG("warning: screening (-T) is ignored\n");
      LOG("    reading initial states: %s\n", o.input.c_str());
      clustering = read_single_column<std::size_t>(o.input);
      if (clustering.size() != n_rows) die("error: initial state file does not match the number of frames");
      read_comments(o.input, cm);
      LOG("    assigning low density states to initial states\n");
      clustering = H::assign_low_density_frames(clustering, hd_idx, fe);
      LOG("    sorting and renaming states by decreasing population\n");
      clustering = H::sorted_cluster_names(clustering);
      LOG("    storing states in: %s\n", o.output.c_str());
      write_clustered_trajectory(o.output, clustering, header, cm);
    } else if (o.has_threshold) {
      LOG("\n~~~ free energy screening\n");
      if (o.threshold.size() > 3)
        die("error: option -T expects at most three floating point arguments: FROM STEP TO.");
      float t_from = 0.1f, t_step = 0.1f;
      float t_to = *std::max_element(fe.begin(), fe.end());
      if (o.threshold.size() >= 1 && o.threshold[0] >= 0.0f) t_from = o.threshold[0];
      if (o.threshold.size() >= 2) t_step = o.threshold[1];
      if (o.threshold.size() == 3) t_to = o.threshold[2];
      auto has2digits = [](float val) {   // density_clustering.cpp:500-504
        const float val_2digits = (int)(val * 100) / 100.0;
        return val_2digits == val;
      };
      if (!(has2digits(t_from) && has2digits(t_step))) die("error: -T can handle at maximum two digits.");
      cm["screening_to"] = t_to;
      cm["screening_from"] = t_from;
      cm["screening_step"] = t_step;
      // one radius graph for max_dist = 4*sigma2 serves every threshold of the scan.  The scan starts
      // from an empty clustering, so a spanning forest with the same connectivity below every
      // threshold does (screening_host.hpp); DC_SCREENING_FULL_GRAPH=1 lists all pairs instead.
      const float max_dist = (float)(4 * sigma2_of(nn_d2));
      const std::vector<H::FreeEnergy> fe_sorted = H::sorted_free_energies(fe);
      H::RadiusGraph graph;
      std::string err;
      const char* full_env = std::getenv("DC_SCREENING_FULL_GRAPH");
      const bool forest = !(full_env && full_env[0] == '1') && n_rows <= ((std::size_t)1 << 24);
      if (forest) {
        if (!H::build_radius_forest(sess(), n_rows, max_dist, fe_sorted, &graph, &err))
          die("error during screening (radius forest)\n" + err);
        LOG("    %zu frame pairs span the graph of the lumping radius\n", graph.n_pairs);
      } else {
        if (!H::build_radius_graph(sess(), n_rows, max_dist, &graph, &err))
          die("error during screening (radius graph)\n" + err);
        LOG("    %zu frame pairs within the lumping radius\n", graph.n_pairs);
      }
      LOG("\n        fe    frames\n");
      // upper limit extended to a 10th of the stepsize to circumvent rounding errors (:796-800)
      const float t_to_low = t_to - t_step / 10.0f + t_step;
      const float t_to_high = t_to + t_step / 10.0f + t_step;
      for (float t = t_from; (t < t_to_low) && !(t_to_high < t); t += t_step) {
        clustering = H::screening_with_graph(fe, fe_sorted, graph, t, clustering);
        std::size_t below = 0;
        for (std::size_t c : clustering) below += (c != 0);
        LOG("    %6.2f %9zu\n", t, below);
        char suffix[32];
        std::snprintf(suffix, sizeof(suffix), ".%0.2f", t);
        write_clustered_trajectory(o.output + suffix, clustering, header, cm);
      }
    } else {
      die("error: one of -T/-i is needed to generate output.");
    }
  }
  LOG("~~~ freeing memory\n");
  dc_hip_session_close(session);
  return EXIT_SUCCESS;
}

}  // namespace

int main(int argc, char** argv) {
  const std::string general_help = std::string("clustering ") + kVersion +
      "\n\nmodes:\n  density: run density clustering (pop / free energy / nearest neighbours on the GPU)\n\n"
      "usage:\n  clustering density --option1 --option2 ...\n\nfor a list of available options:\n"
      "  clustering density -h\n\nthis binary is parallized with HIP (MI355X)\n\n";
  if (argc <= 2) {
    std::cerr << general_help;
    return EXIT_FAILURE;
  }
  const std::string mode(argv[1]);
  if (mode != "density") {
    std::cerr << "\nerror: unrecognized mode '" << mode << "'\n\n"
              << "(this build provides the density mode only; the other modes of moldyn/Clustering\n"
              << " -- network, mpp, coring, noise, filter, stats -- are CPU post-processing tools)\n\n"
              << general_help;
    return EXIT_FAILURE;
  }
  return density_main(argc, argv);
}
