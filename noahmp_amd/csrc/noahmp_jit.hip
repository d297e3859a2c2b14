// Run-time specialisation of the column kernel (opt-in: set_option("jit_option_kernels", 1)).
//
// The kernels of noahmp_engine_d*_r*.hip are the physics compiled with the OPT_* integers as constants -- 22 % faster than the
// generic kernel because the code of every other alternative, and the registers it costs, are gone.  They exist ahead of time
// for five option sets; for any other set this file compiles the same headers once more with hiprtc at the first call that
// brings it (about 20 s, then cached for the life of the process).  Same source, same flags (-O3 -ffp-contract=off), hence the
// same arithmetic; tests compare a run-time compiled kernel with the generic one bit for bit.  Any failure on the way leaves
// the call with the generic kernel.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <dlfcn.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "noahmp_hip.h"
#include "nmp_engine_host.hpp"

namespace nmp_host {

size_t pack_fixed_kargs(const LaunchDesc& d, void* buf, size_t cap);     // noahmp_engine_d1_r1.hip

namespace {

struct JitKernels { hipModule_t mod = nullptr; hipFunction_t fn[2] = {nullptr, nullptr}; bool failed = false; };
std::map<std::string, JitKernels> cache;

std::string source_dir() {                      // the headers live next to this library (noahmp_amd/csrc)
  Dl_info info;
  if (!dladdr((const void*)&pack_fixed_kargs, &info) || !info.dli_fname) return ".";
  std::string p(info.dli_fname);
  const size_t cut = p.rfind('/');
  return cut == std::string::npos ? "." : p.substr(0, cut);
}

const char* kNames[12] = {"DVEG", "CRS", "BTR", "RUN", "SFC", "FRZ", "INF", "RAD", "ALB", "SNF", "TBOT", "STC"};

bool compile(const int* o, JitKernels& out, std::string& log) {
  std::string src =
      "using __hip_internal::int8_t; using __hip_internal::uint8_t; using __hip_internal::int16_t; using __hip_internal::uint16_t;\n"
      "using __hip_internal::int32_t; using __hip_internal::uint32_t; using __hip_internal::int64_t; using __hip_internal::uint64_t;\n";
  for (int i = 0; i < 12; i++) src += std::string("#define NMP_FIXED_") + kNames[i] + " " + std::to_string(o[i]) + "\n";
  src +=
      "#include \"nmp_kernel.hpp\"\n"
      "extern \"C\" __global__ void __launch_bounds__(256, NMP_WAVES_PER_EU) nmp_jit_m0(const nmp::KArgs k) {\n"
      "  column_kernel_body<256, true, 0>(k); }\n"
      "extern \"C\" __global__ void __launch_bounds__(256, NMP_WAVES_PER_EU) nmp_jit_m1(const nmp::KArgs k) {\n"
      "  column_kernel_body<256, true, 1>(k); }\n";
  hiprtcProgram prog = nullptr;
  if (hiprtcCreateProgram(&prog, src.c_str(), "nmp_jit.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS) { log = "hiprtcCreateProgram"; return false; }
  const std::string dir = source_dir();
  const std::string i1 = "-I" + dir, i2 = "-I" + dir + "/../../include";
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", i1.c_str(), i2.c_str()};
  const hiprtcResult r = hiprtcCompileProgram(prog, 6, opts);
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
  std::vector<char> code(n);
  hiprtcGetCode(prog, code.data());
  hiprtcDestroyProgram(&prog);
  if (g.jit_compile_only) { log = "compiled " + std::to_string(n) + " bytes"; return true; }
  if (hipModuleLoadData(&out.mod, code.data()) != hipSuccess) { log = "hipModuleLoadData"; (void)hipGetLastError(); return false; }
  if (hipModuleGetFunction(&out.fn[0], out.mod, "nmp_jit_m0") != hipSuccess ||
      hipModuleGetFunction(&out.fn[1], out.mod, "nmp_jit_m1") != hipSuccess) { log = "hipModuleGetFunction"; (void)hipGetLastError(); return false; }
  return true;
}

}  // namespace

// Launch the kernel specialised for the option values o[12] (DVEG, CRS, BTR, RUN, SFC, FRZ, INF, RAD, ALB, SNF, TBOT, STC), compiling
// it first if this process has not seen the set yet.  mode 0 / 1 as in launch_fixed_*.  Returns false if the caller has to use the
// generic kernel.
bool launch_jit(const int* o, const LaunchDesc& d, int mode, hipStream_t s) {
  std::string key;
  for (int i = 0; i < 12; i++) key += std::to_string(o[i]) + ",";
  auto it = cache.find(key);
  if (it == cache.end()) {
    JitKernels jk;
    std::string log;
    if (!compile(o, jk, log)) { jk.failed = true; g.last_error = "option-specialised kernel not available (" + log + "): generic kernel used"; }
    else if (g.jit_compile_only) { jk.failed = true; g.last_error = log; }
    it = cache.emplace(key, jk).first;
  }
  if (it->second.failed) return false;
  alignas(16) char buf[4096];
  size_t size = pack_fixed_kargs(d, buf, sizeof(buf));
  if (!size) return false;
  const long n = mode == 0 ? (long)d.nti * d.ntj : d.t_count;
  if (n <= 0) return true;
  void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, buf, HIP_LAUNCH_PARAM_BUFFER_SIZE, &size, HIP_LAUNCH_PARAM_END};
  const hipError_t e = hipModuleLaunchKernel(it->second.fn[mode == 1 ? 1 : 0], (unsigned)((n + 255) / 256), 1, 1, 256, 1, 1, 0, s,
                                             nullptr, extra);
  if (e != hipSuccess) { (void)hipGetLastError(); it->second.failed = true; g.last_error = "launch of a run-time compiled kernel failed: generic kernel used"; return false; }
  return true;
}

bool jit_compile_probe(const int* o, std::string& log) {
  const int keep = g.jit_compile_only;
  g.jit_compile_only = 1;
  JitKernels jk;
  const bool ok = compile(o, jk, log);
  g.jit_compile_only = keep;
  return ok;
}

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
