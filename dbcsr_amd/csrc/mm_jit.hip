// mm_jit.hip -- hiprtc front end for mm_exact.h and smm_exact.h (see mm_jit.h).  libhiprtc is opened lazily: a process that never meets a
// mixed-size multiply or a homogeneous parameter stack does not load it.
#include "mm_jit.h"

#include <dlfcn.h>
#include <unistd.h>
#include <cstdint>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "mm_exact.h"   // class_wave_lds(): the same constexpr function the kernel sizes its LDS slice with
#include "smm_exact.h"  // stack_wave_lds()

namespace dbcsr_amd {
namespace {

#include "jit_sources.inc"  // kJitSrc_mm_types, kJitSrc_smm_core, kJitSrc_mm_exact, kJitSrc_smm_exact: the texts of the four headers

typedef struct _hiprtcProgram* rtc_program;
struct Rtc {
  void* lib = nullptr;
  int (*CreateProgram)(rtc_program*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(rtc_program, int, const char**) = nullptr;
  int (*GetProgramLogSize)(rtc_program, size_t*) = nullptr;
  int (*GetProgramLog)(rtc_program, char*) = nullptr;
  int (*GetCodeSize)(rtc_program, size_t*) = nullptr;
  int (*GetCode)(rtc_program, char*) = nullptr;
  int (*DestroyProgram)(rtc_program*) = nullptr;
  int (*Version)(int*, int*) = nullptr;  // (optional)
  bool ok = false;
};

Rtc& rtc() {
  static Rtc r;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"libhiprtc.so", "libhiprtc.so.7", "/opt/rocm/lib/libhiprtc.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (r.lib) break;
    }
    if (!r.lib) {
      fprintf(stderr, "dbcsr_amd: cannot load libhiprtc (%s): mixed block sizes run the generic kernels\n", dlerror());
      return;
    }
#define DBCSR_SYM(field, sym)                               \
  *reinterpret_cast<void**>(&r.field) = dlsym(r.lib, sym);  \
  if (!r.field) return;
    DBCSR_SYM(CreateProgram, "hiprtcCreateProgram")
    DBCSR_SYM(CompileProgram, "hiprtcCompileProgram")
    DBCSR_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
    DBCSR_SYM(GetProgramLog, "hiprtcGetProgramLog")
    DBCSR_SYM(GetCodeSize, "hiprtcGetCodeSize")
    DBCSR_SYM(GetCode, "hiprtcGetCode")
    DBCSR_SYM(DestroyProgram, "hiprtcDestroyProgram")
#undef DBCSR_SYM
    *reinterpret_cast<void**>(&r.Version) = dlsym(r.lib, "hiprtcVersion");
    r.ok = true;
  });
  return r;
}

struct Cached {
  std::vector<char> image;
  hipModule_t mod = nullptr;
  ClassKernel k;
  bool failed = false;
};
std::mutex g_mu;
std::map<std::tuple<int, int, int, int, int, int, int>, Cached> g_cache;  // (device, m, n, k0, k1, k2, g)

bool g_last_from_cache = false;  // (under g_mu) the last compile_and_load found its code object in DBCSR_AMD_JIT_CACHE
// text `defs` (macros + one #include) -> code object -> module; the function `entry` of it.  Caller holds g_mu.
// `image`: the code object's bytes, owned by the caller for as long as the module lives (hipModuleLoadData is not promised to copy them)
int compile_and_load(const char* defs, const char* tu_name, const char* what, const char* entry, const hipDeviceProp_t& prop, hipModule_t* mod,
                     hipFunction_t* fn, size_t* code_size, std::vector<char>* image) {
  Rtc& r = rtc();
  if (!r.ok) return -1;
  g_last_from_cache = false;
  const char* hsrc[] = {kJitSrc_mm_types, kJitSrc_smm_core, kJitSrc_mm_exact, kJitSrc_smm_exact};
  const char* hname[] = {"mm_types.h", "smm_core.h", "mm_exact.h", "smm_exact.h"};
  rtc_program prog = nullptr;
  if (r.CreateProgram(&prog, defs, tu_name, 4, hsrc, hname) != 0) return -1;
  std::string arch = std::string("--offload-arch=") + prop.gcnArchName;
  // DBCSR_AMD_JIT_DEFS: extra -D switches for the kernel text (tuning experiments, see mm_exact.h), space separated
  std::vector<std::string> extra;
  if (const char* d = getenv("DBCSR_AMD_JIT_DEFS")) {
    std::string all(d);
    size_t pos = 0;
    while (pos < all.size()) {
      const size_t e = all.find(' ', pos);
      const std::string tok = all.substr(pos, e == std::string::npos ? std::string::npos : e - pos);
      if (!tok.empty()) extra.push_back(tok);
      if (e == std::string::npos) break;
      pos = e + 1;
    }
  }
  std::vector<const char*> opts = {arch.c_str(), "-O3", "-std=c++17", "-munsafe-fp-atomics"};
  for (const std::string& t : extra) opts.push_back(t.c_str());
  // DBCSR_AMD_JIT_CACHE=<directory>: code objects kept across processes (the reference compiles every triplet again in every run, ~0.5 s each:
  // docs/guide/3-developer-guide/3-programming/2-accelerator-backend/2-libsmm_acc/2-just-in-time-compilation.md).  The file name is a hash of everything the
  // code depends on: the kernel text (macros and the four headers), the options, the architecture, the compiler library's version.
  std::string cache_file;
  if (const char* dir = getenv("DBCSR_AMD_JIT_CACHE")) {
    if (*dir) {
      uint64_t h = 1469598103934665603ull;
      auto mix = [&h](const char* t) {
        for (; *t; ++t) h = (h ^ (unsigned char)*t) * 1099511628211ull;
        h = (h ^ 0xffu) * 1099511628211ull;
      };
      mix(defs);
      for (const char* t : hsrc) mix(t);
      for (const char* t : opts) mix(t);
      int vmaj = 0, vmin = 0;
      if (r.Version) r.Version(&vmaj, &vmin);
      char tail[64];
      snprintf(tail, sizeof tail, "hiprtc %d.%d entry %s", vmaj, vmin, entry);
      mix(tail);
      char name[40];
      snprintf(name, sizeof name, "/dbcsr_amd_%016llx.co", (unsigned long long)h);
      cache_file = std::string(dir) + name;
      if (FILE* f = fopen(cache_file.c_str(), "rb")) {
        std::vector<char>& code = *image;
        code.clear();
        char buf[65536];
        size_t got;
        while ((got = fread(buf, 1, sizeof buf, f)) > 0) code.insert(code.end(), buf, buf + got);
        fclose(f);
        // the file is [magic, payload bytes, payload hash][payload]: a truncated or damaged file must never reach the loader (it throws inside the runtime)
        bool sound = false;
        if (code.size() > 24) {
          uint64_t head[3];
          memcpy(head, code.data(), 24);
          if (head[0] == 0x314f435f444d4143ull && head[1] == code.size() - 24) {
            uint64_t ph = 1469598103934665603ull;
            for (size_t i = 24; i < code.size(); ++i) ph = (ph ^ (unsigned char)code[i]) * 1099511628211ull;
            sound = ph == head[2];
          }
        }
        if (sound) code.erase(code.begin(), code.begin() + 24);
        if (sound && hipModuleLoadData(mod, code.data()) == hipSuccess) {
          if (hipModuleGetFunction(fn, *mod, entry) == hipSuccess) {
            r.DestroyProgram(&prog);
            if (code_size) *code_size = code.size();
            g_last_from_cache = true;
            if (getenv("DBCSR_AMD_MM_VERBOSE")) fprintf(stderr, "dbcsr_amd: %s from %s\n", what, cache_file.c_str());
            return 0;
          }
          (void)hipModuleUnload(*mod);
        }
        (void)hipGetLastError();  // (a truncated or foreign file: compiled again and written over)
      }
    }
  }
  const int rc = r.CompileProgram(prog, (int)opts.size(), opts.data());
  if (rc != 0) {
    size_t ls = 0;
    r.GetProgramLogSize(prog, &ls);
    std::string log(ls + 1, 0);
    if (ls) r.GetProgramLog(prog, &log[0]);
    fprintf(stderr, "dbcsr_amd: hiprtc failed for %s:\n%s\n", what, log.c_str());
    r.DestroyProgram(&prog);
    return -1;
  }
  size_t cs = 0;
  r.GetCodeSize(prog, &cs);
  std::vector<char>& code = *image;
  code.assign(cs, 0);
  r.GetCode(prog, code.data());
  r.DestroyProgram(&prog);
  if (!cache_file.empty()) {  // written beside and renamed: another process never reads half a file
    const std::string tmp = cache_file + "." + std::to_string((long)getpid()) + ".tmp";
    if (FILE* f = fopen(tmp.c_str(), "wb")) {
      uint64_t head[3] = {0x314f435f444d4143ull, (uint64_t)code.size(), 1469598103934665603ull};
      for (char ch : code) head[2] = (head[2] ^ (unsigned char)ch) * 1099511628211ull;
      const bool ok = fwrite(head, 1, 24, f) == 24 && fwrite(code.data(), 1, code.size(), f) == code.size();
      if (fclose(f) != 0 || !ok || rename(tmp.c_str(), cache_file.c_str()) != 0) remove(tmp.c_str());
    }
  }
  if (hipModuleLoadData(mod, code.data()) != hipSuccess) return -1;
  if (hipModuleGetFunction(fn, *mod, entry) != hipSuccess) return -1;
  if (code_size) *code_size = cs;
  return 0;
}

struct CachedStack {
  std::vector<char> image;
  hipModule_t mod = nullptr;
  StackKernel k;
  bool failed = false;
};
std::map<std::tuple<int, int, int, int, int>, CachedStack> g_stack_cache;  // (device, m, n, k, bt)

}  // namespace

int jit_class_kernel(int m, int n, int k0, int k1, int k2, int g, ClassKernel* out) {
  if (g < 1 || g > 64) return -1;
  if (!out || m < 1 || n < 1 || k0 < 1 || m > 32 || n > 32 || k0 > 32 || k1 > 32 || k2 > 32) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(g_mu);
  auto key = std::make_tuple(dev, m, n, k0, k1, k2, g);
  auto it = g_cache.find(key);
  if (it != g_cache.end()) {
    if (it->second.failed) return -1;
    *out = it->second.k;
    return 0;
  }
  Cached& c = g_cache[key];
  c.failed = true;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  const int wave_lds = class_wave_lds(m, n, k0, k1, k2);
  // waves per SIMD the LDS allows (4 waves per workgroup, 160 KiB per CU): the register allocation is asked to allow as many
  int wgs = (160 * 1024) / (4 * wave_lds);
  int minw = wgs > 4 ? 4 : (wgs < 1 ? 1 : wgs);
  if (g > 1 && minw > 1) --minw;  // the multi-block body keeps descriptors and two list windows in registers
  char defs[512];
  snprintf(defs, sizeof defs,
           "#define DBCSR_AMD_JIT_M %d\n#define DBCSR_AMD_JIT_N %d\n#define DBCSR_AMD_JIT_K0 %d\n#define DBCSR_AMD_JIT_K1 %d\n"
           "#define DBCSR_AMD_JIT_K2 %d\n#define DBCSR_AMD_JIT_MINW %d\n#define DBCSR_AMD_JIT_G %d\n#include \"mm_exact.h\"\n",
           m, n, k0, k1, k2, minw, g);
  char what[96];
  snprintf(what, sizeof what, "class (%d, %d; %d, %d, %d)", m, n, k0, k1, k2);
  size_t cs = 0;
  if (compile_and_load(defs, "mm_class.hip", what, "mm_numeric_f64_class", prop, &c.mod, &c.k.fn, &cs, &c.image) != 0) return -1;
  c.k.wave_lds = wave_lds;
  c.failed = false;
  if (getenv("DBCSR_AMD_MM_VERBOSE") && !g_last_from_cache) fprintf(stderr, "dbcsr_amd: compiled class kernel (%d, %d; %d, %d, %d), %zu bytes, %d B LDS per wave\n", m, n, k0, k1, k2, cs, wave_lds);
  *out = c.k;
  return 0;
}


int jit_stack_kernel(int m, int n, int k, bool bt, StackKernel* out, bool compile) {
  if (!out || m < 1 || n < 1 || k < 1 || m > 32 || n > 32 || k > 32) return -1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  std::lock_guard<std::mutex> lock(g_mu);
  auto key = std::make_tuple(dev, m, n, k, (int)bt);
  auto it = g_stack_cache.find(key);
  if (it != g_stack_cache.end()) {
    if (it->second.failed) return -1;
    *out = it->second.k;
    return 0;
  }
  if (!compile) return 1;   // (not there, and the caller does not want to pay for it now)
  CachedStack& c = g_stack_cache[key];
  c.failed = true;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
  const int wave_lds = stack_wave_lds(m, n, k, bt);
  int wgs = (160 * 1024) / (4 * wave_lds);
  const int minw = wgs > 4 ? 4 : (wgs < 1 ? 1 : wgs);
  char defs[512];
  snprintf(defs, sizeof defs,
           "#define DBCSR_AMD_JIT_SM %d\n#define DBCSR_AMD_JIT_SN %d\n#define DBCSR_AMD_JIT_SK %d\n#define DBCSR_AMD_JIT_SBT %d\n"
           "#define DBCSR_AMD_JIT_MINW %d\n#include \"smm_exact.h\"\n",
           m, n, k, bt ? 1 : 0, minw);
  char what[96];
  snprintf(what, sizeof what, "stack kernel (%d, %d, %d%s)", m, n, k, bt ? "; B transposed" : "");
  size_t cs = 0;
  if (compile_and_load(defs, "smm_stack_exact.hip", what, "smm_stack_f64_exact", prop, &c.mod, &c.k.fn, &cs, &c.image) != 0) return -1;
  c.k.wave_lds = wave_lds;
  c.failed = false;
  if (getenv("DBCSR_AMD_MM_VERBOSE") && !g_last_from_cache) fprintf(stderr, "dbcsr_amd: compiled %s, %zu bytes, %d B LDS per wave\n", what, cs, wave_lds);
  *out = c.k;
  return 0;
}

}  // namespace dbcsr_amd
