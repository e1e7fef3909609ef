// Optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg).
#pragma once
#include <hip/hip_runtime.h>

bool ud_prof_on();
void* ud_prof_begin(const char* name, hipStream_t stream);
void ud_prof_end(void* token, hipStream_t stream);

struct UdProfScope {
  void* tok;
  hipStream_t s;
  UdProfScope(const char* name, hipStream_t stream) : tok(nullptr), s(stream) {
    if (ud_prof_on()) tok = ud_prof_begin(name, stream);
  }
  ~UdProfScope() {
    if (tok) ud_prof_end(tok, s);
  }
};
