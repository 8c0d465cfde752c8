#!/usr/bin/env python3
"""Writes /tmp/klat_sweep_prof.hip: the sweep kernel with cycle counters (experiment only).
Per workgroup (first 4096): [0] producer total, [1] wait for memory before the barrier,
[2] barrier, [3] copy-out, [4] nchunks, [5] consumer total, [6] consumer barrier wait."""
s = open('kaldi-lattice-utils_amd/csrc/klat_sweep.hip').read()
def rep(a, b):
    global s
    assert a in s, a[:50]
    s = s.replace(a, b, 1)
rep("namespace klat {\n", "namespace klat {\n__device__ unsigned long long g_prof[8 * 4096];\n#define PROF_T() clock64()\n")
rep("""          if (strict_stores)
            producer_barrier<3 * CPB>();
          else
            producer_barrier<3 * CPB + kIters>();  // this period is staged; the consumer has finished the previous one,
""", """          const long long tb0 = PROF_T();
          if (strict_stores) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CPB) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * CPB + kIters) : "memory");
          const long long tb1 = PROF_T();
          producer_barrier<3 * CPB + kIters>();
          const long long tb2 = PROF_T();
          pr_vm += tb1 - tb0; pr_bar += tb2 - tb1;
""")
rep("""          copy_store();
#pragma unroll
          for (int i = 0; i < CPB; ++i) {""", """          { const long long tc0 = PROF_T(); copy_store(); pr_copy += PROF_T() - tc0; }
#pragma unroll
          for (int i = 0; i < CPB; ++i) {""")
rep("""      static_assert(D % CPB == 0, "prefetch depth must be whole periods");
      for (int c = 0; c < nchunks; c += D) {""", """      long long pr_vm = 0, pr_bar = 0, pr_copy = 0;
      const long long pr_t0 = PROF_T();
      for (int c = 0; c < nchunks; c += D) {""")
rep("""      __syncthreads();  // the consumer has finished the last period
      out_p0 = pend_p0;""", """      if (lane == 0 && blockIdx.x < 4096) {
        unsigned long long* o = g_prof + blockIdx.x * 8;
        o[0] = PROF_T() - pr_t0; o[1] = pr_vm; o[2] = pr_bar; o[3] = pr_copy; o[4] = nchunks;
      }
      __syncthreads();  // the consumer has finished the last period
      out_p0 = pend_p0;""")
rep("""  for (int c = 0; c < nchunks; ++c) {
    if (c % CPB == 0) consumer_barrier();  // the period is staged (and, for c == 0, the ring is initialised)""", """  long long cr_bar = 0;
  const long long cr_t0 = PROF_T();
  for (int c = 0; c < nchunks; ++c) {
    if (c % CPB == 0) { const long long a0 = PROF_T(); consumer_barrier(); cr_bar += PROF_T() - a0; }""")
rep("""  __syncthreads();  // lets the producer copy out the last period's states
  if (FAST) {""", """  if (lane == 0 && blockIdx.x < 4096) { g_prof[blockIdx.x * 8 + 5] = PROF_T() - cr_t0; g_prof[blockIdx.x * 8 + 6] = cr_bar; }
  __syncthreads();  // lets the producer copy out the last period's states
  if (FAST) {""")
rep("""hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream) {""", """extern "C" void klat_debug_sweep_profile(unsigned long long* out, int n) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * (size_t)n);
}
hipError_t launch_sweep(const SweepArgs& args, int num_jobs, int G, hipStream_t stream) {""")
open('/tmp/klat_sweep_prof.hip', 'w').write(s)
