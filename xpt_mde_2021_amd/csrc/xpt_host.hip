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
