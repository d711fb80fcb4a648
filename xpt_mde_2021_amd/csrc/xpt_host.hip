// xpt_host.hip -- host-side helpers of the input contract (no device code).
// xpt_crc32c: CRC-32C (Castagnoli) as used by the TFRecord framing the reference's input pipeline reads
// (tfrecords/tfrecord_reader.py:61-75 -> tf.data.TFRecordDataset): each record is
//   uint64 length | uint32 masked_crc32c(length) | payload | uint32 masked_crc32c(payload).
// Hardware CRC32 instruction (SSE4.2) when the build host has it, slicing-by-1 table otherwise.
#include <stddef.h>
#include <stdint.h>

#include "../../include/xpt_hip.h"

namespace {

uint32_t table_[256];
bool table_ready_ = false;

void build_table() {
  for (uint32_t i = 0; i < 256; ++i) {
    uint32_t c = i;
    for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : (c >> 1);
    table_[i] = c;
  }
  table_ready_ = true;
}

#if defined(__x86_64__)
__attribute__((target("sse4.2"))) uint32_t crc_hw(uint32_t crc, const unsigned char* p, size_t n) {
  uint64_t c = crc;
  while (n >= 8) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    c = __builtin_ia32_crc32di(c, v);
    p += 8;
    n -= 8;
  }
  uint32_t c32 = (uint32_t)c;
  while (n--) c32 = __builtin_ia32_crc32qi(c32, *p++);
  return c32;
}
#endif

uint32_t crc_sw(uint32_t crc, const unsigned char* p, size_t n) {
  if (!table_ready_) build_table();
  while (n--) crc = table_[(crc ^ *p++) & 0xFF] ^ (crc >> 8);
  return crc;
}

}  // namespace

extern "C" uint32_t xpt_crc32c(const void* data, size_t nbytes) {
  if (data == nullptr || nbytes == 0) return 0;
  const unsigned char* p = (const unsigned char*)data;
  uint32_t crc = 0xFFFFFFFFu;
#if defined(__x86_64__)
  if (__builtin_cpu_supports("sse4.2")) return crc_hw(crc, p, nbytes) ^ 0xFFFFFFFFu;
#endif
  return crc_sw(crc, p, nbytes) ^ 0xFFFFFFFFu;
}

// ---------------------------------------------------------------------------------------------- hipGraph node census
// A captured training step must not contain memset nodes: on this runtime a hipMemsetAsync captured into a hipGraph
// writes garbage from the second replay on (tools/replay_probe_memset.py).  The trainers walk every
// captured graph with this helper (train_val._StepGraph) and refuse one that has such a node, instead of relying on the
// convention "no memset inside the step".  counts[6] = kernel, memcpy, memset, host, other (empty / event / ...) nodes and
// the total, child graphs included.
#include <hip/hip_runtime.h>
#include <vector>

namespace {
int census_walk(hipGraph_t graph, int* counts, int depth) {
  size_t n = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) return XPT_ERR_LAUNCH;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) return XPT_ERR_LAUNCH;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) return XPT_ERR_LAUNCH;
    if (t == hipGraphNodeTypeGraph && depth < 8) {
      hipGraph_t child;
      if (hipGraphChildGraphNodeGetGraph(nodes[i], &child) != hipSuccess) return XPT_ERR_LAUNCH;
      const int rc = census_walk(child, counts, depth + 1);
      if (rc != XPT_OK) return rc;
      continue;
    }
    const int slot = t == hipGraphNodeTypeKernel ? 0 : t == hipGraphNodeTypeMemcpy ? 1 : t == hipGraphNodeTypeMemset ? 2
                     : t == hipGraphNodeTypeHost ? 3 : 4;
    counts[slot] += 1;
    counts[5] += 1;
  }
  return XPT_OK;
}
}  // namespace

extern "C" int xpt_graph_node_census(void* hip_graph, int* counts) {
  if (hip_graph == nullptr || counts == nullptr) return XPT_ERR_NULL;
  for (int i = 0; i < 6; ++i) counts[i] = 0;
  (void)hipGetLastError();
  return census_walk((hipGraph_t)hip_graph, counts, 0);
}

/* image-to-XCD affinity of the kernels that follow the convention of xpt_common.h (xpt_xcd_remap): 0 = off */
int g_xpt_xcd_affinity = 1;
extern "C" int xpt_set_xcd_affinity(int on) {
  g_xpt_xcd_affinity = on ? 1 : 0;
  return XPT_OK;
}
