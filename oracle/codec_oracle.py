"""CPU restatement of gr-lora's encode and decode blocks (PDU <-> bytes).  TEST INFRASTRUCTURE ONLY.

Pure-Python loops that follow the reference statement by statement (packets are a few hundred symbols):
  decode: lib/decode_impl.cc:133-162 (gray, whiten), :177-275 (deinterleave), :279-372 (hamming_decode), :420-513 (decode)
  encode: lib/encode_impl.cc:131-162 (gray, whiten), :177-262 (interleave), :264-292 (hamming_encode), :318-424 (encode)
Only tests/ may import this.  The whitening sequence is an argument: the reference's tables (include/lora/lora.h:29-46)
are reference source and are never stored in this repository; tests that want them parse that header at test time where
/root/reference exists.  PARITY UNPINNED: the reference holds no vectors for these blocks either (python/qa_decode.py).
"""
WHITENING_SEQUENCE_LENGTH = 279          # include/lora/lora.h:27
# lib/decode_impl.cc:30-33
D_P1, D_P2, D_P4, D_P8 = 0xAA, 0x66, 0x1E, 0xFE
# lib/encode_impl.cc:28-31
E_P1, E_P2, E_P4, E_P8 = 0x0D, 0x0B, 0x07, 0xFF


def _parity(c, bitmask):                  # decode_impl.cc:374-387 / encode_impl.cc:294-307
    shiftme = c & bitmask & 0xFF
    parity = 0
    for _ in range(8):
        if shiftme & 0x1:
            parity += 1
        shiftme >>= 1
    return parity % 2


def to_gray(symbols):                     # decode_impl.cc:133-140
    return [((s >> 1) ^ s) & 0xFFFF for s in symbols]


def from_gray(symbols):                   # encode_impl.cc:142-153 (unsigned short arithmetic: >> 16 is 0)
    out = []
    for s in symbols:
        s = s ^ (s >> 16); s = s ^ (s >> 8); s = s ^ (s >> 4); s = s ^ (s >> 2); s = s ^ (s >> 1)
        out.append(s & 0xFFFF)
    return out


def deinterleave(symbols, ppm, rdd):      # decode_impl.cc:177-275
    symbols = list(symbols)
    codewords = []
    for i in range(len(symbols)):         # :191-197 swap the two top bits
        s = symbols[i]
        symbols[i] = ((s & (0x1 << (ppm - 1))) >> 1 | (s & (0x1 << (ppm - 2))) << 1 | (s & ((0x1 << (ppm - 2)) - 1))) & 0xFFFF
    for block_count in range(len(symbols) // (4 + rdd)):
        block = [0] * 12
        bit_idx = 0
        bit_offset = 0
        for bitcount in range(ppm * (4 + rdd)):
            if symbols[(bitcount % (4 + rdd)) + (4 + rdd) * block_count] & ((0x1 << (ppm - 1)) >> ((bit_idx + bit_offset) % ppm)):
                block[bitcount // (4 + rdd)] |= 0x1 << (bitcount % (4 + rdd))
                block[bitcount // (4 + rdd)] &= 0xFF
            if bitcount % (4 + rdd) == (4 + rdd - 1):
                bit_idx = 0
                bit_offset += 1
            else:
                bit_idx += 1
        for cw_idx in range(ppm):
            b = block[cw_idx]
            if rdd == 4:
                b = (b & 128) | (b & 64) | (b & 32) >> 5 | (b & 16) | (b & 8) << 2 | (b & 4) << 1 | (b & 2) << 1 | (b & 1) << 1
            elif rdd == 3:
                b = (b & 64) | (b & 32) | (b & 16) >> 1 | (b & 8) << 1 | (b & 4) | (b & 2) | (b & 1)
            block[cw_idx] = (b & 0xFF) & ((1 << (4 + rdd)) - 1)
        if ppm == 8:
            codewords += [block[6], block[7], block[4], block[5]]
        elif ppm == 7:
            codewords += [block[6], block[4], block[5]]
        elif ppm == 6:
            codewords += [block[4], block[5]]
        elif ppm == 5:
            codewords += [block[4]]
        codewords += [block[2], block[3], block[0], block[1]]
    return codewords


def hamming_decode(codewords, rdd):       # decode_impl.cc:279-372
    out = []
    for cw in codewords:
        p1 = p2 = p4 = 0
        if rdd >= 3:
            p4 = _parity(cw, D_P4 >> (4 - rdd))
        if rdd >= 2:
            p2 = _parity(cw, D_P2 >> (4 - rdd))
        if rdd >= 1:
            p1 = _parity(cw, D_P1 >> (4 - rdd))
        error_pos = -1
        if p1 != 0: error_pos += 1
        if p2 != 0: error_pos += 2
        if p4 != 0: error_pos += 4
        if rdd > 2:
            num_set_bits = bin(cw & 0xFF).count("1")
            if error_pos >= 0 and num_set_bits < 6 and num_set_bits > 2:
                cw ^= (0x80 >> (4 - rdd)) >> error_pos
        if rdd in (1, 2):
            cw = cw & 0x0F
        elif rdd == 3:
            cw = (((cw & 0x10) >> 1) | (cw & 0x04) | (cw & 0x02) | (cw & 0x01)) & 0x0F
        elif rdd == 4:
            cw = (((cw & 0x20) >> 2) | ((cw & 0x08) >> 1) | ((cw & 0x04) >> 1) | ((cw & 0x02) >> 1)) & 0x0F
        out.append(cw & 0x0F)
    return out


def decode_pdu(sf, cr, ldr, whitening, symbols):          # decode_impl.cc:420-513
    s = to_gray(list(symbols))
    for i in range(min(len(s), WHITENING_SEQUENCE_LENGTH, len(whitening) if whitening is not None else 0)):
        s[i] = s[i] ^ whitening[i]
    header, payload = s[:8], s[8:]
    nyb = hamming_decode(deinterleave(header, sf - 2, 4), 4)
    nyb += hamming_decode(deinterleave(payload, (sf - 2) if ldr else sf, cr), cr)
    out = []
    for i, v in enumerate(nyb):
        if i % 2 == 0:
            out.append((v << 4) & 0xF0)
        else:
            out[i // 2] |= v & 0x0F
    return bytes(out)


def hamming_encode(nybbles, rdd):         # encode_impl.cc:264-292 (rdd is unused there too: always the (8,4) codeword)
    out = []
    for n in nybbles:
        p1 = _parity(n, E_P1); p2 = _parity(n, E_P2); p4 = _parity(n, E_P4)
        p8 = _parity(n | p1 << 7 | p2 << 6 | p4 << 4, E_P8)
        out.append(((p1 << 7) | (p2 << 6) | (p8 << 5) | (p4 << 4) | (n & 0x08) | (n & 0x04) | (n & 0x02) | (n & 0x01)) & 0xFF)
    return out


def interleave(codewords, ppm, rdd):      # encode_impl.cc:177-262; only ppm 6 and 8 fill `reordered` (:207-226)
    if ppm not in (6, 8):
        raise ValueError("the reference's interleaver leaves its block uninitialised unless ppm is 6 or 8")
    symbols = []
    for block_count in range(len(codewords) // ppm):
        block = [0] * 8
        reordered = [0] * 8
        c = codewords[block_count * ppm:(block_count + 1) * ppm]
        if ppm == 6:
            reordered[4], reordered[5], reordered[2], reordered[3], reordered[0], reordered[1] = c
        else:
            reordered[0], reordered[7], reordered[2], reordered[1], reordered[4], reordered[3], reordered[6], reordered[5] = c
        bit_idx = 0
        bit_offset = 0
        for bitcount in range(ppm * (4 + rdd)):
            if reordered[bitcount // (4 + rdd)] & (0x1 << (bitcount % (4 + rdd))):
                block[bitcount % (4 + rdd)] |= ((0x1 << (ppm - 1)) >> ((bit_idx + bit_offset) % ppm))
            if bitcount % (4 + rdd) == (4 + rdd - 1):
                bit_idx = 0
                bit_offset += 1
            else:
                bit_idx += 1
        symbols += [b & 0xFF for b in block[:4 + rdd]]
    return [((s & (0x1 << (ppm - 1))) >> 1 | (s & (0x1 << (ppm - 2))) << 1 | (s & ((0x1 << (ppm - 2)) - 1))) & 0xFFFF for s in symbols]


def encode_pdu(sf, cr, ldr, whitening, data, header=False):     # encode_impl.cc:318-424
    nybbles = [0xF] * 8 if header else []
    for b in data:
        nybbles += [(b & 0xF0) >> 4, b & 0x0F]
    head, pay = nybbles[:sf - 2], nybbles[sf - 2:]
    symbols = interleave(hamming_encode(head, 4), sf - 2, 4) + interleave(hamming_encode(pay, cr), (sf - 2) if ldr else sf, cr)
    for i in range(min(len(symbols), WHITENING_SEQUENCE_LENGTH, len(whitening) if whitening is not None else 0)):
        symbols[i] = ((symbols[i] & 0xFF) ^ whitening[i]) & 0xFF          # :158: the encoder whitens 8-bit symbols only
    symbols = from_gray(symbols)
    ldr_limit = len(symbols) if ldr else 8
    return [(s << 2) & 0xFFFF if i < ldr_limit else s for i, s in enumerate(symbols)]
