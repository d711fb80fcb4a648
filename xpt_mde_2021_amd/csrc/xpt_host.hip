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

/* 16-bit activation format this library was built with (xpt_common.h): 0 = bfloat16 (libxpt_hip.so), 1 = IEEE half (libxpt_hip_f16.so) */
extern "C" int xpt_half_format(void) {
#ifdef XPT_HALF_F16
  return 1;
#else
  return 0;
#endif
}

/* image-to-XCD affinity of the kernels that follow the convention of xpt_common.h (xpt_xcd_remap): 0 = off */
int g_xpt_xcd_affinity = 1;
extern "C" int xpt_set_xcd_affinity(int on) {
  g_xpt_xcd_affinity = on ? 1 : 0;
  return XPT_OK;
}

// ---------------------------------------------------------------------------------------------- TFRecord shards, natively
// Round 4: the reader's per-record work in C (the Python version parsed the protobuf wire format byte by byte under the
// GIL and sustained 1,578 snippets/s -- below the 1,730 images/s training step it feeds).  Same contract as
// tfrecords/tfrecord_reader.py:61-108 + tfr_util.py:8-77 of the reference: TFRecord framing with masked CRC-32C, payload =
// serialized tf.train.Example, one bytes_list (ndarray.tobytes()) or int64_list value per key.
namespace {

inline uint32_t mask_crc(uint32_t crc) { return ((crc >> 15) | (crc << 17)) + 0xA282EAD8u; }

inline bool read_varint(const unsigned char* p, size_t n, size_t& pos, uint64_t& out) {
  uint64_t v = 0;
  int shift = 0;
  while (pos < n && shift < 64) {
    const unsigned char b = p[pos++];
    v |= (uint64_t)(b & 0x7F) << shift;
    if (!(b & 0x80)) {
      out = v;
      return true;
    }
    shift += 7;
  }
  return false;
}

// one protobuf field of the message [p, p + n) at pos: field number, wire type, varint value or (offset, length) of its bytes
struct Field {
  uint32_t number, wire;
  uint64_t value;
  size_t off, len;
};
inline bool next_field(const unsigned char* p, size_t n, size_t& pos, Field& f) {
  uint64_t tag;
  if (!read_varint(p, n, pos, tag)) return false;
  f.number = (uint32_t)(tag >> 3);
  f.wire = (uint32_t)(tag & 7);
  f.value = 0; f.off = 0; f.len = 0;
  if (f.wire == 0) return read_varint(p, n, pos, f.value);
  if (f.wire == 2) {
    uint64_t ln;
    if (!read_varint(p, n, pos, ln) || ln > n - pos) return false;
    f.off = pos; f.len = (size_t)ln;
    pos += (size_t)ln;
    return true;
  }
  const size_t w = f.wire == 5 ? 4 : f.wire == 1 ? 8 : 0;
  if (!w || w > n - pos) return false;
  f.off = pos; f.len = w;
  pos += w;
  return true;
}

}  // namespace

/* Frames of one shard held in memory (a read-only mapping): payload_off[k] / payload_len[k] / payload_crc[k] of record k
 * (the stored masked CRC of the payload, checked later by xpt_tfrecord_decode on the decoding thread).  verify_crc: the
 * 12-byte headers are checked here.  Returns the number of records (<= max_records: call again with larger arrays when it
 * returns max_records), or -(k + 1) when record k is truncated or its header CRC does not match. */
extern "C" long long xpt_tfrecord_index(const void* shard, size_t nbytes, int verify_crc, unsigned long long* payload_off,
                                        unsigned long long* payload_len, unsigned int* payload_crc, long long max_records) {
  if (shard == nullptr || payload_off == nullptr || payload_len == nullptr || payload_crc == nullptr) return XPT_ERR_NULL;
  const unsigned char* p = (const unsigned char*)shard;
  size_t pos = 0;
  long long k = 0;
  while (pos + 12 <= nbytes && k < max_records) {
    uint64_t len;
    uint32_t hcrc;
    __builtin_memcpy(&len, p + pos, 8);
    __builtin_memcpy(&hcrc, p + pos + 8, 4);
    if (verify_crc && hcrc != mask_crc(xpt_crc32c(p + pos, 8))) return -(k + 1);
    if (len > nbytes || pos + 12 + len + 4 > nbytes) return -(k + 1);
    payload_off[k] = pos + 12;
    payload_len[k] = len;
    uint32_t pcrc;
    __builtin_memcpy(&pcrc, p + pos + 12 + len, 4);
    payload_crc[k] = pcrc;
    pos += 12 + (size_t)len + 4;
    ++k;
  }
  return k;
}

/* One record -> the rows of a batch's staging buffers: for key i (NUL-terminated UTF-8 name keys[i]) the first bytes_list
 * value is copied to dst[i] (exactly dst_bytes[i] bytes, else the record is rejected), an int64_list value is stored as one
 * int64 (dst_bytes[i] = 8), a float_list value as one float (4).  Returns 0; -10 CRC mismatch; -11 malformed Example;
 * -(100 + i) key i missing; -(1000 + i) key i has the wrong size.  Thread-safe, no allocation. */
extern "C" int xpt_tfrecord_decode(const void* payload, size_t nbytes, unsigned int stored_crc, int verify_crc, int nkeys,
                                   const char* const* keys, void* const* dst, const size_t* dst_bytes) {
  if (payload == nullptr || keys == nullptr || dst == nullptr || dst_bytes == nullptr) return XPT_ERR_NULL;
  if (nkeys <= 0 || nkeys > 64) return XPT_ERR_ARG;
  const unsigned char* p = (const unsigned char*)payload;
  if (verify_crc && stored_crc != mask_crc(xpt_crc32c(p, nbytes))) return -10;
  size_t klen[64];
  bool found[64];
  for (int i = 0; i < nkeys; ++i) {
    klen[i] = __builtin_strlen(keys[i]);
    found[i] = false;
  }
  size_t pos = 0;
  Field ex;
  while (pos < nbytes) {
    if (!next_field(p, nbytes, pos, ex)) return -11;
    if (ex.number != 1 || ex.wire != 2) continue;                         // Example.features
    const unsigned char* fp = p + ex.off;
    size_t fpos = 0;
    Field entry;
    while (fpos < ex.len) {
      if (!next_field(fp, ex.len, fpos, entry)) return -11;
      if (entry.number != 1 || entry.wire != 2) continue;                 // Features.feature map entry
      const unsigned char* ep = fp + entry.off;
      size_t epos = 0;
      Field kv, key{}, feat{};
      bool has_key = false, has_feat = false;
      while (epos < entry.len) {
        if (!next_field(ep, entry.len, epos, kv)) return -11;
        if (kv.number == 1 && kv.wire == 2) { key = kv; has_key = true; }
        else if (kv.number == 2 && kv.wire == 2) { feat = kv; has_feat = true; }
      }
      if (!has_key || !has_feat) continue;
      int which = -1;
      for (int i = 0; i < nkeys; ++i)
        if (klen[i] == key.len && __builtin_memcmp(keys[i], ep + key.off, key.len) == 0) { which = i; break; }
      if (which < 0) continue;                                            // a feature the side-car config does not list
      const unsigned char* tp = ep + feat.off;                            // Feature: oneof bytes_list = 1 / float_list = 2 / int64_list = 3
      size_t tpos = 0;
      Field kind;
      while (tpos < feat.len) {
        if (!next_field(tp, feat.len, tpos, kind)) return -11;
        if (kind.wire != 2) continue;
        const unsigned char* lp = tp + kind.off;
        size_t lpos = 0;
        Field val;
        while (lpos < kind.len) {
          if (!next_field(lp, kind.len, lpos, val)) return -11;
          if (val.number != 1) continue;
          if (kind.number == 1) {                                         // bytes_list.value[0]
            if (val.wire != 2 || val.len != dst_bytes[which]) return -(1000 + which);
            __builtin_memcpy(dst[which], lp + val.off, val.len);
          } else if (kind.number == 3) {                                  // int64_list.value[0] (packed or not)
            uint64_t v = val.value;
            if (val.wire == 2) {
              size_t q = 0;
              if (!read_varint(lp + val.off, val.len, q, v)) return -11;
            }
            if (dst_bytes[which] != 8) return -(1000 + which);
            const int64_t sv = (int64_t)v;
            __builtin_memcpy(dst[which], &sv, 8);
          } else if (kind.number == 2) {                                  // float_list.value[0] (packed or not)
            if (dst_bytes[which] != 4 || val.len < 4) return -(1000 + which);
            __builtin_memcpy(dst[which], lp + val.off, 4);
          }
          found[which] = true;
          break;                                                          // first value only
        }
      }
    }
  }
  for (int i = 0; i < nkeys; ++i)
    if (!found[i]) return -(100 + i);
  return 0;
}
