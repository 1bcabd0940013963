// decode_chain.cpp -- the PDU decode chain of gr-lora's decode block as a plain C ABI (include/lora_decode.h).
// Host-side integer code (a few hundred symbols per packet): gray -> de-whiten -> de-interleave -> Hamming -> bytes.
// Restated from the behaviour of lib/decode_impl.cc (cited per function), organised per interleaver block with the
// diagonal pattern in closed form; the whitening sequence comes from the caller.  Never calls anything under oracle/.
#include "lora_decode.h"

#include <vector>

namespace {

// decode_impl::deinterleave, lib/decode_impl.cc:177-275.  In: n symbols of ppm bits; out: codewords of 4 + rdd bits.
// Bit b of codeword c of a block comes from symbol b of the block, bit ppm-1 - ((b + c) mod ppm), after the two top bits
// of every symbol have been swapped (:191-197); a block is 4 + rdd symbols and yields ppm codewords, of which the
// reference keeps at most eight, in the order of :243-270.  Symbols beyond the last whole block are dropped (:200).
void deinterleave(const std::vector<uint16_t> &symbols, unsigned ppm, unsigned rdd, std::vector<uint8_t> &codewords) {
  const unsigned width = 4 + rdd;
  const unsigned top = 1u << (ppm - 1), second = 1u << (ppm - 2);
  for (size_t blk = 0; blk < symbols.size() / width; blk++) {
    uint8_t block[12] = {0};                                   // INTERLEAVER_BLOCK_SIZE, :35
    for (unsigned b = 0; b < width; b++) {
      const unsigned s0 = symbols[blk * width + b];
      const unsigned s = ((s0 & top) >> 1) | ((s0 & second) << 1) | (s0 & (second - 1));
      for (unsigned c = 0; c < ppm; c++)
        if (s & (top >> ((b + c) % ppm))) block[c] |= (uint8_t)(1u << b);
    }
    for (unsigned c = 0; c < ppm; c++) {
      unsigned v = block[c];
      if (rdd == 4)        // traditional Hamming order, :226-228
        v = (v & 128) | (v & 64) | ((v & 32) >> 5) | (v & 16) | ((v & 8) << 2) | ((v & 4) << 1) | ((v & 2) << 1) | ((v & 1) << 1);
      else if (rdd == 3)   // :230-232
        v = (v & 64) | (v & 32) | ((v & 16) >> 1) | ((v & 8) << 1) | (v & 4) | (v & 2) | (v & 1);
      block[c] = (uint8_t)(v & ((1u << width) - 1));
    }
    static const int order8[] = {6, 7, 4, 5}, order7[] = {6, 4, 5}, order6[] = {4, 5}, order5[] = {4};
    const int *lead = nullptr; int n_lead = 0;
    if (ppm == 8) { lead = order8; n_lead = 4; }
    else if (ppm == 7) { lead = order7; n_lead = 3; }
    else if (ppm == 6) { lead = order6; n_lead = 2; }
    else if (ppm == 5) { lead = order5; n_lead = 1; }           // ppm > 8 (SF9-SF12 payload): only the four below, :243-270
    for (int i = 0; i < n_lead; i++) codewords.push_back(block[lead[i]]);
    codewords.push_back(block[2]); codewords.push_back(block[3]); codewords.push_back(block[0]); codewords.push_back(block[1]);
  }
}

inline unsigned parity_of(unsigned c, unsigned mask) { return (unsigned)__builtin_popcount(c & mask & 0xFFu) & 1u; }   // :374-387

// decode_impl::hamming_decode, lib/decode_impl.cc:279-372: one nybble per codeword
void hamming_decode(const std::vector<uint8_t> &codewords, unsigned rdd, std::vector<uint8_t> &nybbles) {
  for (unsigned cw : codewords) {
    if (rdd > 2) {
      // syndrome from the three parity checks, masks of :30-32 shifted for the shorter code (:295-300)
      const unsigned p4 = parity_of(cw, 0x1Eu >> (4 - rdd)), p2 = parity_of(cw, 0x66u >> (4 - rdd)),
                     p1 = parity_of(cw, 0xAAu >> (4 - rdd));
      const int error_pos = -1 + (int)p1 + 2 * (int)p2 + 4 * (int)p4;                 // :304-307
      const unsigned set_bits = (unsigned)__builtin_popcount(cw & 0xFFu);
      if (error_pos >= 0 && set_bits < 6 && set_bits > 2) cw ^= (0x80u >> (4 - rdd)) >> error_pos;   // :324-327
      cw &= 0xFFu;
    }
    unsigned d;
    switch (rdd) {                                               // data bits, :350-369
      case 3:  d = ((cw & 0x10) >> 1) | (cw & 0x04) | (cw & 0x02) | (cw & 0x01); break;
      case 4:  d = ((cw & 0x20) >> 2) | ((cw & 0x08) >> 1) | ((cw & 0x04) >> 1) | ((cw & 0x02) >> 1); break;
      default: d = cw; break;
    }
    nybbles.push_back((uint8_t)(d & 0x0F));
  }
}

}  // namespace

extern "C" {

size_t lora_decode_max_bytes(size_t n_symbols) {
  // at most 8 codewords (nybbles) per block of >= 5 symbols, header block included; two nybbles per byte
  return (n_symbols / 5 + 2) * 4 + 1;
}

int lora_decode_pdu(unsigned sf, unsigned code_rate, int low_data_rate, const uint16_t *whitening, size_t n_whitening,
                    const uint16_t *symbols, size_t n_symbols, uint8_t *bytes_out, size_t cap, size_t *n_bytes) {
  if (sf < 6 || sf > 12 || code_rate < 1 || code_rate > 4) return LORA_DECODE_EINVAL;     // asserts at :62-64
  if ((n_symbols && !symbols) || (n_whitening && !whitening) || !n_bytes) return LORA_DECODE_EINVAL;
  try {
    std::vector<uint16_t> header, payload;
    for (size_t i = 0; i < n_symbols; i++) {
      unsigned s = symbols[i];
      s = ((s >> 1) ^ s) & 0xFFFFu;                                                        // to_gray, :133-140
      if (i < n_whitening && i < LORA_WHITENING_SEQUENCE_LENGTH) s ^= whitening[i];        // whiten, :155-162
      (i < 8 ? header : payload).push_back((uint16_t)s);                                   // :444-445
    }
    std::vector<uint8_t> cw, nyb;
    deinterleave(header, sf - 2, 4, cw);                                                   // :460, always ppm = sf-2, rdd = 4
    hamming_decode(cw, 4, nyb);                                                            // :466
    cw.clear();
    deinterleave(payload, low_data_rate ? sf - 2 : sf, code_rate, cw);                     // :470
    hamming_decode(cw, code_rate, nyb);                                                    // :476
    const size_t nb = (nyb.size() + 1) / 2;                                                // :485-497: high nybble first
    *n_bytes = nb;
    if (nb > cap || (nb && !bytes_out)) return LORA_DECODE_ENOSPC;
    for (size_t i = 0; i < nb; i++) bytes_out[i] = 0;
    for (size_t i = 0; i < nyb.size(); i++) bytes_out[i / 2] |= (i % 2 == 0) ? (uint8_t)((nyb[i] << 4) & 0xF0) : (uint8_t)(nyb[i] & 0x0F);
    return LORA_DECODE_OK;
  } catch (...) { return LORA_DECODE_EINVAL; }
}

}  // extern "C"
