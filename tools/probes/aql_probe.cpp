// Probe: does an AQL dispatch chain with acquire/release fence scope NONE keep the XCD L2s warm
// between dependent kernels (barrier bit set), and what does a kernel boundary cost then?
// Pure HSA (no HIP runtime): own queue, own code object, hand-written packets.
// Build: g++ -O2 -std=c++17 aql_probe.cpp -I/opt/rocm/include -L/opt/rocm/lib -lhsa-runtime64 -o aql_probe
// Run:   ./aql_probe aql_probe_kernel.hsaco
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; \
  hsa_status_string(s_, &m_); fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, m_ ? m_ : "?"); exit(2); } } while (0)

static hsa_agent_t g_gpu, g_cpu; static bool have_gpu = false, have_cpu = false;
static hsa_amd_memory_pool_t g_dev_pool, g_karg_pool; static bool have_dev = false, have_karg = false;

static hsa_status_t agent_cb(hsa_agent_t a, void*) {
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !have_gpu) { g_gpu = a; have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !have_cpu) { g_cpu = a; have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t dev_pool_cb(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  bool alloc; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !have_dev) { g_dev_pool = p; have_dev = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t cpu_pool_cb(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !have_karg) { g_karg_pool = p; have_karg = true; }
  return HSA_STATUS_SUCCESS;
}

struct Kernel { uint64_t object; uint32_t karg, group, priv; };

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "aql_probe_kernel.hsaco";
  setvbuf(stdout, nullptr, _IONBF, 0);
  CK(hsa_init());
  CK(hsa_iterate_agents(agent_cb, nullptr));
  if (!have_gpu || !have_cpu) { fprintf(stderr, "no agents\n"); return 2; }
  char name[64]; hsa_agent_get_info(g_gpu, HSA_AGENT_INFO_NAME, name); printf("gpu agent: %s\n", name);
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, dev_pool_cb, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(g_cpu, cpu_pool_cb, nullptr));
  if (!have_dev || !have_karg) { fprintf(stderr, "no pools\n"); return 2; }

  // code object
  FILE* f = fopen(path, "rb"); if (!f) { perror(path); return 2; }
  fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<char> blob(sz); if (fread(blob.data(), 1, sz, f) != (size_t)sz) return 2; fclose(f);
  hsa_code_object_reader_t rd; CK(hsa_code_object_reader_create_from_memory(blob.data(), sz, &rd));
  hsa_executable_t ex; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
  CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
  CK(hsa_executable_freeze(ex, nullptr));
  hsa_executable_symbol_t sym; CK(hsa_executable_get_symbol_by_name(ex, "read_panels.kd", &g_gpu, &sym));
  Kernel k;
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.karg));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.group));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.priv));
  printf("kernel object %llx kernarg %u group %u private %u\n", (unsigned long long)k.object, k.karg, k.group, k.priv);

  hsa_queue_t* q; CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
  const bool prof = getenv("AQL_PROBE_PROFILING") != nullptr;
  if (prof) CK(hsa_amd_profiling_set_profiler_enabled(q, 1));

  const size_t nbuf = 2048 * 4352, nout = 256 * 512;
  double *buf, *out;
  CK(hsa_amd_memory_pool_allocate(g_dev_pool, nbuf * 8, 0, (void**)&buf));
  CK(hsa_amd_memory_pool_allocate(g_dev_pool, nout * 8, 0, (void**)&out));
  CK(hsa_amd_memory_fill(buf, 0, nbuf * 2));   // count in uint32 words
  CK(hsa_amd_memory_fill(out, 0, nout * 2));

  struct Args { const double* buf; double* out; int mode; unsigned int target; unsigned int* counters; };
  unsigned int* counters; CK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096, 0, (void**)&counters));
  CK(hsa_amd_memory_fill(counters, 0, 1024));
  const int NARG = 64;
  std::vector<int> slot_mode;   // kernarg slot -> mode word
  char* hk; CK(hsa_amd_memory_pool_allocate(g_karg_pool, 4096 * NARG, 0, (void**)&hk));
  CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, hk));
  const int modes[] = {2, 2 | (3 << 4), 2 | (1 << 12), 2 | (2 << 12), 2 | (3 << 12),
                       2 | (1 << 12) | (1 << 14), 2 | (3 << 12) | (1 << 14), 2 | (1 << 12) | (2 << 14), 2 | (3 << 12) | (2 << 14)};
  for (int m = 0; m < NARG; ++m) {
    memset(hk + 4096 * m, 0, 4096);
    const int md = m < int(sizeof modes / sizeof modes[0]) ? modes[m] : 2 | (3 << 4);
    slot_mode.push_back(md);
    Args a{buf, out, md, 0, counters}; memcpy(hk + 4096 * m, &a, sizeof a);
  }
  auto slot_of = [&](int md) { for (int m = 0; m < NARG; ++m) if (slot_mode[m] == md) return m; fprintf(stderr, "no slot for mode %d\n", md); exit(2); return 0; };
  char* kargs = hk;
  if (!getenv("AQL_PROBE_HOST_KERNARG")) {   // kernargs in device memory (what HIP does on this family)
    CK(hsa_amd_memory_pool_allocate(g_dev_pool, 4096 * NARG, 0, (void**)&kargs));
    CK(hsa_memory_copy(kargs, hk, 4096 * NARG));
  }
  printf("kernargs in %s memory\n", kargs == hk ? "host" : "device");

  const int CH = 200;
  std::vector<hsa_signal_t> sig(CH);
  for (auto& s : sig) CK(hsa_signal_create(1, 0, nullptr, &s));

  // chain without per-packet signals: device time from the first and last packet's stamps
  auto chain = [&](int mode, int acq, int rel, int n, const char* label) {
    hsa_signal_store_relaxed(sig[0], 1); hsa_signal_store_relaxed(sig[1], 1);
    const uint32_t mask = q->size - 1;
    int done = 0;
    auto t0 = std::chrono::steady_clock::now();
    while (done < n) {
      const int m = (n - done) < 2048 ? (n - done) : 2048;
      // wait for room
      while (hsa_queue_load_write_index_relaxed(q) + m - hsa_queue_load_read_index_scacquire(q) > q->size) {}
      uint64_t idx0 = hsa_queue_add_write_index_relaxed(q, m);
      for (int i = 0; i < m; ++i) {
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((idx0 + i) & mask);
        p->setup = 1; p->workgroup_size_x = 512; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = 256 * 512; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = k.priv; p->group_segment_size = k.group;
        p->kernel_object = k.object; p->kernarg_address = kargs + 4096 * slot_of(mode); p->reserved2 = 0;
        const bool first = (done + i == 0), last = (done + i == n - 1);
        p->completion_signal = first ? sig[0] : last ? sig[1] : hsa_signal_t{0};
        uint16_t hdr = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                       (((first || last) ? HSA_FENCE_SCOPE_SYSTEM : acq) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                       (((first || last) ? HSA_FENCE_SCOPE_SYSTEM : rel) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint16_t*)p, hdr, __ATOMIC_RELEASE);
      }
      hsa_signal_store_screlease(q->doorbell_signal, idx0 + m - 1);
      done += m;
    }
    while (hsa_signal_wait_scacquire(sig[1], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) {}
    auto t1 = std::chrono::steady_clock::now();
    if (label && !prof)
      printf("%-72s chain of %d: %.2f us/dispatch host wall\n", label, n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    if (label && prof) {
      uint64_t tf; hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &tf);
      hsa_amd_profiling_dispatch_time_t a, b;
      CK(hsa_amd_profiling_get_dispatch_time(g_gpu, sig[0], &a)); CK(hsa_amd_profiling_get_dispatch_time(g_gpu, sig[1], &b));
      printf("%-72s chain of %d: %.2f us/dispatch device, %.2f host wall\n", label, n,
             double(b.end - a.start) * 1e6 / double(tf) / n, std::chrono::duration<double, std::micro>(t1 - t0).count() / n);
    }
  };

  auto run = [&](int mode, int acq, int rel, bool barrier, const char* label) {
    for (auto& s : sig) hsa_signal_store_relaxed(s, 1);
    const uint32_t mask = q->size - 1;
    uint64_t idx0 = hsa_queue_add_write_index_relaxed(q, CH);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < CH; ++i) {
      hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((idx0 + i) & mask);
      p->setup = 1;   // 1 dimension
      p->workgroup_size_x = 512; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
      p->grid_size_x = 256 * 512; p->grid_size_y = 1; p->grid_size_z = 1;
      p->private_segment_size = k.priv; p->group_segment_size = k.group;
      p->kernel_object = k.object; p->kernarg_address = kargs + 4096 * slot_of(mode); p->reserved2 = 0;
      p->completion_signal = sig[i];
      const bool first_last = (i == 0 || i == CH - 1);
      uint16_t hdr = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                     ((barrier ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                     ((first_last ? HSA_FENCE_SCOPE_SYSTEM : acq) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                     ((first_last ? HSA_FENCE_SCOPE_SYSTEM : rel) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
      __atomic_store_n((uint16_t*)p, hdr, __ATOMIC_RELEASE);
    }
    hsa_signal_store_screlease(q->doorbell_signal, idx0 + CH - 1);
    while (hsa_signal_wait_scacquire(sig[CH - 1], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) {}
    auto t1 = std::chrono::steady_clock::now();
    uint64_t tf; hsa_system_get_info(HSA_SYSTEM_INFO_TIMESTAMP_FREQUENCY, &tf);
    double dur = 0, gap = 0; uint64_t prev_end = 0, first_start = 0, last_end = 0; int n = 0;
    for (int i = 0; i < CH; ++i) {
      hsa_amd_profiling_dispatch_time_t t; CK(hsa_amd_profiling_get_dispatch_time(g_gpu, sig[i], &t));
      if (i == 0) first_start = t.start;
      if (i >= 10 && i < CH - 1) { dur += double(t.end - t.start); gap += double(t.start) - double(prev_end); ++n; }
      prev_end = t.end; last_end = t.end;
    }
    const double us = 1e6 / double(tf);
    printf("%-46s kernel %.2f us  gap %.2f us  chain %.2f us/dispatch (host wall %.2f)\n", label,
           dur / n * us, gap / n * us, double(last_end - first_start) * us / CH,
           std::chrono::duration<double, std::micro>(t1 - t0).count() / CH);
  };

  const int AG = HSA_FENCE_SCOPE_AGENT, NO = HSA_FENCE_SCOPE_NONE;
  chain(2, AG, AG, 60000, "warm-up");
  // ---- overlapped chain: per-dispatch kernarg slots (64 B apart) carrying the arrival target --------------------
  const int OC = 3000;
  char* ok_host; CK(hsa_amd_memory_pool_allocate(g_karg_pool, 64 * OC, 0, (void**)&ok_host));
  CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, ok_host));
  char* ok_dev; CK(hsa_amd_memory_pool_allocate(g_dev_pool, 64 * OC, 0, (void**)&ok_dev));
  unsigned int launches_so_far = 0;   // synced launches since the counters were zeroed
  auto overlapped = [&](int mode, bool barrier, bool sync, uint32_t lds_bytes, const char* label) {
    double best = 1e30, worst = 0;
    for (int rep = 0; rep < 5; ++rep) {
      for (int i = 0; i < OC; ++i) {
        Args a{buf, out, mode | (sync ? (1 << 16) : 0), 256u * (launches_so_far + i), counters};
        memset(ok_host + 64 * i, 0, 64); memcpy(ok_host + 64 * i, &a, sizeof a);
      }
      CK(hsa_memory_copy(ok_dev, ok_host, 64 * OC));
      hsa_signal_store_relaxed(sig[0], 1); hsa_signal_store_relaxed(sig[1], 1);
      const uint32_t mask = q->size - 1;
      auto t0 = std::chrono::steady_clock::now();
      uint64_t idx0 = hsa_queue_add_write_index_relaxed(q, OC);
      for (int i = 0; i < OC; ++i) {
        hsa_kernel_dispatch_packet_t* p = (hsa_kernel_dispatch_packet_t*)q->base_address + ((idx0 + i) & mask);
        p->setup = 1; p->workgroup_size_x = 512; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = 256 * 512; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = k.priv; p->group_segment_size = k.group + lds_bytes;
        p->kernel_object = k.object; p->kernarg_address = ok_dev + 64 * i; p->reserved2 = 0;
        const bool first = (i == 0), last = (i == OC - 1);
        p->completion_signal = last ? sig[1] : hsa_signal_t{0};
        uint16_t hdr = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) |
                       (((barrier || first || last) ? 1 : 0) << HSA_PACKET_HEADER_BARRIER) |
                       (((first || last) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                       (((first || last) ? HSA_FENCE_SCOPE_SYSTEM : HSA_FENCE_SCOPE_AGENT) << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n((uint16_t*)p, hdr, __ATOMIC_RELEASE);
      }
      hsa_signal_store_screlease(q->doorbell_signal, idx0 + OC - 1);
      while (hsa_signal_wait_scacquire(sig[1], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0) {}
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / OC;
      if (sync) launches_so_far += OC;
      if (us < best) best = us; if (us > worst) worst = us;
    }
    double flag = 0; CK(hsa_memory_copy(&flag, out, 8));
    printf("%-84s min %.2f max %.2f us/dispatch%s\n", label, best, worst, flag < 0 ? "  [SPIN TIMEOUT]" : "");
  };
  if (getenv("AQL_PROBE_OVERLAP")) {
    chain(2, AG, AG, 60000, "warm-up");
    const int PK = 2 | (3 << 12) | (1 << 14);   // packed blocks, one group ahead
    const int EMPTY = 2 | (3 << 4);
    for (uint32_t lds : {0u, 100u * 1024u}) {
      char lab[160];
      snprintf(lab, sizeof lab, "LDS %3u KB: empty kernel, barrier bit", lds >> 10); overlapped(EMPTY, true, false, lds, lab);
      snprintf(lab, sizeof lab, "LDS %3u KB: empty kernel, no barrier bit, no sync", lds >> 10); overlapped(EMPTY, false, false, lds, lab);
      snprintf(lab, sizeof lab, "LDS %3u KB: empty kernel, no barrier bit, counter sync", lds >> 10); overlapped(EMPTY, false, true, lds, lab);
      snprintf(lab, sizeof lab, "LDS %3u KB: packed operand reads, barrier bit", lds >> 10); overlapped(PK, true, false, lds, lab);
      snprintf(lab, sizeof lab, "LDS %3u KB: packed operand reads, barrier bit + counter sync", lds >> 10); overlapped(PK, true, true, lds, lab);
      snprintf(lab, sizeof lab, "LDS %3u KB: packed operand reads, no barrier bit, no sync", lds >> 10); overlapped(PK, false, false, lds, lab);
      snprintf(lab, sizeof lab, "LDS %3u KB: packed operand reads, no barrier bit, counter sync", lds >> 10); overlapped(PK, false, true, lds, lab);
    }
    printf("done\n");
    return 0;
  }

  auto best_of = [&](int mode, int acq, int rel, const char* lab) {
    double best = 1e30, worst = 0;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      chain(mode, acq, rel, 3000, nullptr);
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 3000;
      if (us < best) best = us; if (us > worst) worst = us;
    }
    printf("%-84s min %.2f max %.2f us/dispatch\n", lab, best, worst);
  };
  best_of(2 | (3 << 4), AG, AG, "empty kernel, agent/agent");
  best_of(2 | (3 << 4), NO, NO, "empty kernel, none/none");
  best_of(2, AG, AG, "contiguous panel per wave, 34 loads in flight, agent/agent");
  best_of(2 | (1 << 12), AG, AG, "fragment shape, all 32 loads in flight, agent/agent");
  best_of(2 | (1 << 12), NO, NO, "fragment shape, all 32 loads in flight, none/none");
  best_of(2 | (2 << 12), AG, AG, "fragment shape, one group (8 loads) at a time, agent/agent");
  best_of(2 | (3 << 12), AG, AG, "fragment shape, one group ahead (16 in flight), agent/agent");
  best_of(2 | (3 << 12), NO, NO, "fragment shape, one group ahead (16 in flight), none/none");
  best_of(2 | (1 << 12) | (1 << 14), AG, AG, "packed blocks (32-byte lane stride), all 32 in flight, agent/agent");
  best_of(2 | (3 << 12) | (1 << 14), AG, AG, "packed blocks (32-byte lane stride), one group ahead, agent/agent");
  best_of(2 | (1 << 12) | (2 << 14), AG, AG, "packed slabs (1 KB per instruction), all 32 in flight, agent/agent");
  best_of(2 | (3 << 12) | (2 << 14), AG, AG, "packed slabs (1 KB per instruction), one group ahead, agent/agent");
  printf("done\n");
  return 0;
}
