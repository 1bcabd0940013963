/*
 * lora_decode.h -- C ABI of the PDU decode chain behind the demod block.
 *
 * Host code, no GPU: what gr-lora's decode block does to one demodulated symbol PDU
 * (reference: lib/decode_impl.cc:420-513, decode_impl::decode):
 *     gray mapping (:133-140) -> de-whitening (:155-162) -> split into 8 header symbols and the rest (:444-445)
 *     -> de-interleaving (:177-275; header at ppm = sf - 2, rdd = 4, payload at ppm = sf or sf - 2, rdd = code rate)
 *     -> Hamming(4 + rdd, 4) decoding (:279-372) -> nybbles packed into bytes (:485-497).
 * The whitening sequence is an argument: the reference's sequences (include/lora/lora.h:29-46) were derived
 * empirically by its authors and are reference source, so they are not part of this library; a caller that links
 * gr-lora passes its own table (e.g. gr::lora::whitening_sequence_sf8_implicit, 279 entries used).
 */
#ifndef LORA_DECODE_H
#define LORA_DECODE_H

#include <stddef.h>
#include <stdint.h>

#if defined(__GNUC__)
#define LORA_DECODE_API __attribute__((visibility("default")))
#else
#define LORA_DECODE_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* whitening_sequence_length, include/lora/lora.h:27: the reference whitens at most this many symbols */
#define LORA_WHITENING_SEQUENCE_LENGTH 279

enum { LORA_DECODE_OK = 0, LORA_DECODE_EINVAL = -1, LORA_DECODE_ENOSPC = -2 };

/* Upper bound of the bytes one PDU of n_symbols decodes to (for sizing bytes_out). */
LORA_DECODE_API size_t lora_decode_max_bytes(size_t n_symbols);

/*
 * decode_impl::decode for one PDU (the u16vector the demod block publishes, lib/demod_impl.cc:420-422).
 *   sf 6..12, code_rate 1..4 (parity bits, as decode::make, lib/decode_impl.cc:43-46), low_data_rate 0/1.
 *   whitening / n_whitening: the sequence and how many of its entries may be used; the reference uses
 *   min(n_symbols, LORA_WHITENING_SEQUENCE_LENGTH) entries (:157).  NULL with n_whitening 0 = no de-whitening.
 *   bytes_out / cap: output buffer; *n_bytes receives the count (LORA_DECODE_ENOSPC if cap is too small).
 * Explicit-header mode is "not yet supported" by the reference (:115-119: it only prints a warning and decodes as
 * implicit), so there is no header argument.
 */
LORA_DECODE_API int lora_decode_pdu(unsigned sf, unsigned code_rate, int low_data_rate,
                                    const uint16_t *whitening, size_t n_whitening,
                                    const uint16_t *symbols, size_t n_symbols,
                                    uint8_t *bytes_out, size_t cap, size_t *n_bytes);

#ifdef __cplusplus
}
#endif
#endif /* LORA_DECODE_H */
