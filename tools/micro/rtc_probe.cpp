// rtc_probe.cpp -- does in-process hipRTC work on the GPU box?  compile a step_fast instance, load it, read its attributes.
#include <hip/hip_runtime.h>
#include <hip/hiprtc.h>
#include <cstdlib>
#include <cstdio>
#include <vector>
#include <string>
#include <chrono>
#include <fstream>
#include <sstream>
static std::string slurp(const char* p){ std::ifstream f(p); std::stringstream s; s<<f.rdbuf(); return s.str(); }
int main(int argc, char** argv){
  std::string R = std::string(getenv("GRAFT_REPO_ROOT") ? getenv("GRAFT_REPO_ROOT") : "/root/repo") + "/";
  std::string hdr = slurp((R+"include/sgw.h").c_str());
  // strip #include <stdint.h>
  size_t q = hdr.find("#include <stdint.h>"); hdr.replace(q, 19, "");
  std::string src = R"(
typedef unsigned char uint8_t; typedef signed char int8_t; typedef unsigned short uint16_t; typedef short int16_t;
typedef unsigned int uint32_t; typedef int int32_t; typedef unsigned long long uint64_t; typedef long long int64_t;
typedef unsigned long uintptr_t;
#define offsetof(t, m) __builtin_offsetof(t, m)
)";
  src += hdr;
  for (const char* n : {"common.h","step_generic.h","step_fast.h","step_big.h","phase.h","small_kernels.h"}) {
    std::string s = slurp((R+"sorrel_amd/csrc/"+n).c_str());
    size_t k; while ((k = s.find("#pragma once")) != std::string::npos) s.replace(k, 12, "");
    src += s;
  }
  hiprtcProgram prog;
  hiprtcCreateProgram(&prog, src.c_str(), "sgw_jit.hip", 0, nullptr, nullptr);
  for (int i = 1; i < argc; ++i) hiprtcAddNameExpression(prog, argv[i]);
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17"};
  auto t0 = std::chrono::steady_clock::now();
  hiprtcResult r = hiprtcCompileProgram(prog, 3, opts);
  auto t1 = std::chrono::steady_clock::now();
  size_t ls; hiprtcGetProgramLogSize(prog, &ls); std::string log(ls, 0); hiprtcGetProgramLog(prog, log.data());
  printf("rc=%d %s %.2fs\nlog=%.3000s\n", r, hiprtcGetErrorString(r), std::chrono::duration<double>(t1-t0).count(), log.c_str());
  for (int i = 1; i < argc; ++i) { const char* ln=nullptr; hiprtcGetLoweredName(prog, argv[i], &ln); printf("lowered %s\n", ln?ln:"(null)"); }
  size_t cs=0; hiprtcGetCodeSize(prog, &cs); printf("code %zu\n", cs);
  std::vector<char> code(cs); hiprtcGetCode(prog, code.data());
  hipModule_t mod; hipError_t e = hipModuleLoadData(&mod, code.data()); printf("load: %s\n", hipGetErrorString(e));
  for (int i = 1; i < argc && e == hipSuccess; ++i) { const char* ln=nullptr; hiprtcGetLoweredName(prog, argv[i], &ln); hipFunction_t f; e = hipModuleGetFunction(&f, mod, ln); printf("func: %s\n", hipGetErrorString(e));
    int v=0; hipFuncGetAttribute(&v, HIP_FUNC_ATTRIBUTE_NUM_REGS, f); printf("regs %d\n", v); }
}
