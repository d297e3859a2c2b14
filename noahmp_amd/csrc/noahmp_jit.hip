// Run-time specialisation of the column kernel (default on; set_option("jit_option_kernels", 0) switches it off).
//
// The kernels of noahmp_engine_d*_r*.hip are the physics compiled with the OPT_* integers (noahmp_options, lsm:9352-9388; the
// twelve arguments of drv:15-17) as constants -- 22 % faster than the generic kernel because the code of every other alternative,
// and the registers it costs, are gone.  They exist ahead of time for five option sets; for any other set this file compiles the
// same headers once more with hiprtc at the first call that brings it (3-10 s) and keeps the code object on disk, keyed by the
// twelve options, a hash of every source file and build macro that goes into the kernel, the hiprtc version and the GPU
// architecture -- so only the first process that ever meets an option set pays for the compilation.  Cache directory:
// $NOAHMP_HIP_CACHE_DIR, else jit_cache/ next to this library if it is writable, else ~/.cache/noahmp_hip.  Same source, same
// flags (-O3 -ffp-contract=off), same build macros, hence the same arithmetic; tests compare a run-time compiled kernel with the
// generic one bit for bit.  Any failure on the way leaves the call with the generic kernel and says so once on stderr.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dirent.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "noahmp_hip.h"
#include "nmp_engine_host.hpp"

namespace nmp_host {

size_t pack_fixed_kargs(const LaunchDesc& d, void* buf, size_t cap);     // noahmp_engine_d1_r1.hip

namespace {

struct JitKernels { hipModule_t mod = nullptr; hipFunction_t fn[3] = {nullptr, nullptr, nullptr}; bool failed = false; };
std::map<std::string, JitKernels> cache;
struct JitStats { int compiled = 0, disk_hits = 0, fallbacks = 0; } stats;

std::string source_dir() {                      // the headers live next to this library (noahmp_amd/csrc)
  Dl_info info;
  if (!dladdr((const void*)&pack_fixed_kargs, &info) || !info.dli_fname) return ".";
  std::string p(info.dli_fname);
  const size_t cut = p.rfind('/');
  return cut == std::string::npos ? "." : p.substr(0, cut);
}

const char* kNames[12] = {"DVEG", "CRS", "BTR", "RUN", "SFC", "FRZ", "INF", "RAD", "ALB", "SNF", "TBOT", "STC"};

#define NMP_STR2(x) #x
#define NMP_STR(x) NMP_STR2(x)
// the build macros that change the kernel's arithmetic or resources: the run-time compiled kernels get this library's values
#ifndef NMP_EXACT_LIBM
#define NMP_EXACT_LIBM 1
#endif
#ifndef NMP_WAVES_PER_EU
#define NMP_WAVES_PER_EU 2
#endif
#ifndef NMP_LIBM_LDS
#define NMP_LIBM_LDS 1
#endif
// Experiment builds (build(extra_flags=...)) define further macros the kernel headers read: they travel too, so that a run-time
// compiled kernel never runs other physics than this library's ahead-of-time ones.
const char* kBuildMacros =
    "#define NMP_EXACT_LIBM " NMP_STR(NMP_EXACT_LIBM) "\n"
    "#define NMP_WAVES_PER_EU " NMP_STR(NMP_WAVES_PER_EU) "\n"
    "#define NMP_LIBM_LDS " NMP_STR(NMP_LIBM_LDS) "\n"
    "#define NMP_FIXED_BLOCK " NMP_STR(NMP_FIXED_BLOCK) "\n"
    "#define NOAHMP_NSOIL " NMP_STR(NOAHMP_NSOIL) "\n"
#ifdef NMP_PHASE_TIMERS
    "#define NMP_PHASE_TIMERS 1\n"
#endif
    ;
// the wrapper around the headers (kernel names, launch bounds): part of the cache key like the headers themselves
const char* kWrapper =
    "#include \"nmp_kernel.hpp\"\n"
    "extern \"C\" __global__ void __launch_bounds__(NMP_FIXED_BLOCK, NMP_WAVES_PER_EU) nmp_jit_m0(const nmp::KArgs k) {\n"
    "  __shared__ float lds[nmp::LAY_SLOTS * NMP_FIXED_BLOCK];\n"
    "  column_kernel_body<NMP_FIXED_BLOCK, true, 0>(k, lds, (long)blockIdx.x, k.t_first, k.t_count); }\n"
    "extern \"C\" __global__ void __launch_bounds__(NMP_FIXED_BLOCK, NMP_WAVES_PER_EU) nmp_jit_m4(const nmp::KArgs k) {\n"      // = noahmp_ranges_kernel
    "  __shared__ float lds[nmp::LAY_SLOTS * NMP_FIXED_BLOCK];\n"
    "  const long nb_ice = (k.r_ice + NMP_FIXED_BLOCK - 1) / NMP_FIXED_BLOCK, nb_land = (k.r_land + NMP_FIXED_BLOCK - 1) / NMP_FIXED_BLOCK;\n"
    "  const long b = (long)blockIdx.x;\n"
    "  if (b < nb_ice) column_kernel_body<NMP_FIXED_BLOCK, true, 2>(k, lds, b, k.r_land, k.r_ice);\n"
    "  else if (b < nb_ice + nb_land) column_kernel_body<NMP_FIXED_BLOCK, true, 1>(k, lds, b - nb_ice, 0L, k.r_land);\n"
    "  else column_kernel_body<NMP_FIXED_BLOCK, true, 3>(k, lds, b - nb_ice - nb_land, k.r_land + k.r_ice, k.r_skip); }\n";
// (the scheduler strategy: as noahmp_amd/build.py -- fewer hazard s_nop in the issue-bound column kernel, round 5)
// (-instcombine-max-copied-from-constant-users: without it the one-launch class-range kernel copies its 1.6 KB argument block to scratch, build.py)
const char* kCompileFlags[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                               "-mllvm", "-instcombine-max-copied-from-constant-users=100000"};
constexpr int kNumCompileFlags = sizeof(kCompileFlags) / sizeof(kCompileFlags[0]);

uint64_t fnv1a(const void* p, size_t n, uint64_t h = 1469598103934665603ull) {
  const unsigned char* c = (const unsigned char*)p;
  for (size_t i = 0; i < n; i++) { h ^= c[i]; h *= 1099511628211ull; }
  return h;
}

bool read_file(const std::string& path, std::vector<char>& out) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  out.resize(n > 0 ? n : 0);
  const bool ok = n > 0 && fread(out.data(), 1, n, f) == (size_t)n;
  fclose(f);
  return ok;
}

// hash of everything the kernel is compiled from: every header next to the library (sorted by name), the C-ABI header, the
// build macros, the compiler flags and the hiprtc version.  0 = the sources are not there (no run-time compilation possible).
uint64_t source_hash() {
  static uint64_t h = 0;
  static bool done = false;
  if (done) return h;
  done = true;
  const std::string dir = source_dir();
  std::vector<std::string> files;
  if (DIR* d = opendir(dir.c_str())) {
    while (dirent* e = readdir(d)) {
      const std::string n = e->d_name;
      if ((n.size() > 4 && n.substr(n.size() - 4) == ".hpp") || (n.size() > 4 && n.substr(n.size() - 4) == ".inc")) files.push_back(dir + "/" + n);
    }
    closedir(d);
  }
  std::sort(files.begin(), files.end());
  files.push_back(dir + "/../../include/noahmp_hip.h");
  uint64_t acc = fnv1a(kBuildMacros, strlen(kBuildMacros));
  acc = fnv1a(kWrapper, strlen(kWrapper), acc);
  for (const char* f : kCompileFlags) acc = fnv1a(f, strlen(f), acc);
  int major = 0, minor = 0;
  hiprtcVersion(&major, &minor);
  acc = fnv1a(&major, sizeof major, acc);
  acc = fnv1a(&minor, sizeof minor, acc);
  bool kernel_seen = false;
  std::vector<char> buf;
  for (const std::string& f : files) {
    if (!read_file(f, buf)) { if (f.find("noahmp_hip.h") != std::string::npos) return h = 0; continue; }
    if (f.find("nmp_kernel.hpp") != std::string::npos) kernel_seen = true;
    acc = fnv1a(buf.data(), buf.size(), acc);
  }
  return h = kernel_seen ? (acc ? acc : 1) : 0;
}

bool writable_dir(const std::string& d) {
  mkdir(d.c_str(), 0755);
  return access(d.c_str(), W_OK | X_OK) == 0;
}

std::string cache_dir() {
  static std::string dir;
  static bool done = false;
  if (done) return dir;
  done = true;
  if (const char* e = getenv("NOAHMP_HIP_CACHE_DIR")) { if (*e && writable_dir(e)) return dir = e; }
  const std::string local = source_dir() + "/jit_cache";
  if (writable_dir(local)) return dir = local;
  if (const char* home = getenv("HOME")) {
    const std::string c = std::string(home) + "/.cache";
    mkdir(c.c_str(), 0755);
    if (writable_dir(c + "/noahmp_hip")) return dir = c + "/noahmp_hip";
  }
  return dir = "";
}

std::string cache_file(const int* o) {
  const std::string d = cache_dir();
  const uint64_t h = source_hash();
  if (d.empty() || !h) return "";
  char name[160];
  snprintf(name, sizeof name, "/nmp_gfx950_o%d_%d_%d_%d_%d_%d_%d_%d_%d_%d_%d_%d_%016llx.hsaco", o[0], o[1], o[2], o[3], o[4], o[5], o[6],
           o[7], o[8], o[9], o[10], o[11], (unsigned long long)h);
  return d + name;
}

void store_code(const std::string& path, const std::vector<char>& code) {
  if (path.empty()) return;
  char tmp[64];
  snprintf(tmp, sizeof tmp, ".tmp%d", (int)getpid());
  const std::string t = path + tmp;
  FILE* f = fopen(t.c_str(), "wb");
  if (!f) return;
  const bool ok = fwrite(code.data(), 1, code.size(), f) == code.size();
  fclose(f);
  if (ok) rename(t.c_str(), path.c_str());        // atomic: concurrent ranks may compile the same set
  else unlink(t.c_str());
}

bool load_module(const std::vector<char>& code, JitKernels& out, std::string& log) {
  if (hipModuleLoadData(&out.mod, code.data()) != hipSuccess) { log = "hipModuleLoadData"; (void)hipGetLastError(); return false; }
  if (hipModuleGetFunction(&out.fn[0], out.mod, "nmp_jit_m0") != hipSuccess ||
      hipModuleGetFunction(&out.fn[1], out.mod, "nmp_jit_m4") != hipSuccess) { log = "hipModuleGetFunction"; (void)hipGetLastError(); return false; }
  return true;
}

bool compile(const int* o, JitKernels& out, std::string& log) {
  const std::string cpath = cache_file(o);
  std::vector<char> code;
  if (!cpath.empty() && read_file(cpath, code)) {            // another process (or an earlier run) compiled this set already
    stats.disk_hits++;
    if (g.jit_compile_only) { log = "cached " + std::to_string(code.size()) + " bytes"; return true; }
    if (load_module(code, out, log)) return true;
    code.clear();                                            // unreadable / stale object: compile again
    out = JitKernels();
  }
  if (!source_hash()) { log = "kernel sources not found next to the library (" + source_dir() + ")"; return false; }
  std::string src =
      "using __hip_internal::int8_t; using __hip_internal::uint8_t; using __hip_internal::int16_t; using __hip_internal::uint16_t;\n"
      "using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;\n";
  src += kBuildMacros;
  for (int i = 0; i < 12; i++) src += std::string("#define NMP_FIXED_") + kNames[i] + " " + std::to_string(o[i]) + "\n";
  src += kWrapper;
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "nmp_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram"; return false; }
  const std::string dir = source_dir();
  const std::string i1 = "-I" + dir, i2 = "-I" + dir + "/../../include";
  std::vector<const char*> opts(kCompileFlags, kCompileFlags + kNumCompileFlags);
  opts.push_back(i1.c_str()); opts.push_back(i2.c_str());
  const hiprtcResult r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    hiprtcGetProgramLogSize(prog, &n);
    std::vector<char> buf(n + 1, 0);
    if (n) hiprtcGetProgramLog(prog, buf.data());
    log = std::string("hiprtcCompileProgram: ") + buf.data();
    hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t n = 0;
  hiprtcGetCodeSize(prog, &n);
  code.resize(n);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  stats.compiled++;
  store_code(cpath, code);
  if (g.jit_compile_only) { log = "compiled " + std::to_string(n) + " bytes"; return true; }
  return load_module(code, out, log);
}

}  // namespace

// Launch the kernel specialised for the option values o[12] (DVEG, CRS, BTR, RUN, SFC, FRZ, INF, RAD, ALB, SNF, TBOT, STC), compiling
// it first if this process has not seen the set yet.  mode 0 / 4 and the start / stop events as in launch_fixed_*.  Returns false if the
// caller has to use the generic kernel.
bool launch_jit(const int* o, const LaunchDesc& d, int mode, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
  std::string key;
  for (int i = 0; i < 12; i++) key += std::to_string(o[i]) + ",";
  auto it = cache.find(key);
  if (it == cache.end()) {
    JitKernels jk;
    std::string log;
    if (!compile(o, jk, log)) {
      jk.failed = true;
      g.last_error = "option-specialised kernel not available (" + log + "): generic kernel used";
      if (!stats.fallbacks++) fprintf(stderr, "noahmp_hip: %s\n", g.last_error.c_str());      // say it once, then only in last_error
    }
    else if (g.jit_compile_only) { jk.failed = true; g.last_error = log; }
    it = cache.emplace(key, jk).first;
  }
  if (it->second.failed) return false;
  alignas(16) char buf[4096];
  size_t size = pack_fixed_kargs(d, buf, sizeof(buf));
  if (!size) return false;
  constexpr long B = NMP_FIXED_BLOCK;
  const long nb = mode == 4 ? (d.r_ice + B - 1) / B + (d.r_land + B - 1) / B + (d.r_skip + B - 1) / B : ((long)d.nti * d.ntj + B - 1) / B;
  if (nb <= 0) return true;
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  // (hipExtModuleLaunchKernel takes the grid in work-items)
  const hipError_t e = hipExtModuleLaunchKernel(it->second.fn[mode == 4 ? 1 : 0], (uint32_t)(nb * B), 1, 1, NMP_FIXED_BLOCK, 1, 1, 0, s, nullptr, extra, ev0, ev1, 0);
  if (e != hipSuccess) { (void)hipGetLastError(); it->second.failed = true; g.last_error = "launch of a run-time compiled kernel failed: generic kernel used"; return false; }
  return true;
}

std::string cache_dir_public() { return cache_dir(); }
unsigned long long source_hash_public() { return source_hash(); }

bool jit_compile_probe(const int* o, std::string& log) {
  const int keep = g.jit_compile_only;
  g.jit_compile_only = 1;
  JitKernels jk;
  const bool ok = compile(o, jk, log);
  g.jit_compile_only = keep;
  return ok;
}

// out[0] = option sets compiled by this process, out[1] = loaded from the on-disk cache, out[2] = fell back to the generic kernel
void jit_stats(int* out) { out[0] = stats.compiled; out[1] = stats.disk_hits; out[2] = stats.fallbacks; }

void jit_finalize() {
  for (auto& kv : cache) if (kv.second.mod) hipModuleUnload(kv.second.mod);
  cache.clear();
}

}  // namespace nmp_host

// Self-test without a GPU: compile the specialised kernels for an option set (no load, no launch).  0 = compiled; the compiler's
// message (or "compiled N bytes") is copied to log.
extern "C" int noahmp_hip_jit_compile_check(const int32_t* options12, char* log, size_t cap) {
  int o[12];
  for (int i = 0; i < 12; i++) o[i] = options12[i];
  std::string msg;
  const bool ok = nmp_host::jit_compile_probe(o, msg);
  if (log && cap) { strncpy(log, msg.c_str(), cap - 1); log[cap - 1] = 0; }
  return ok ? 0 : 1;
}

// the hash that keys the cache files (kernel sources + build macros + compiler flags + hiprtc version); 0 = sources not found
extern "C" unsigned long long noahmp_hip_jit_source_hash(void) { return nmp_host::source_hash_public(); }

// {compiled by this process, loaded from the on-disk cache, fell back to the generic kernel}; returns the cache directory ("" = none)
extern "C" const char* noahmp_hip_jit_cache_info(int32_t* counts3) {
  static std::string dir;
  dir = nmp_host::cache_dir_public();
  if (counts3) { int c[3]; nmp_host::jit_stats(c); for (int i = 0; i < 3; i++) counts3[i] = c[i]; }
  return dir.c_str();
}
