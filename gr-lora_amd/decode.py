"""Python mirror of the reference's decode block for the PDU chain behind the demod.

    lora.decode(spreading_factor, code_rate, low_data_rate, header)       grc/lora_decode.xml:7
    gr::lora::decode::make(short, short, bool, bool)                      include/lora/decode.h, lib/decode_impl.cc:43-51

Message port "in" takes the demod's symbol PDUs, "out" publishes cons(make_dict(), u8vector) (lib/decode_impl.cc:59-61,
:499,:511-512); here work(symbols) returns the bytes.  The numeric work is the C ABI of include/lora_decode.h (host
code).  The whitening sequence is an argument (reference: include/lora/lora.h:29-46, chosen by sf and low_data_rate at
lib/decode_impl.cc:72-113): those tables are reference source and are not part of this package.
"""
import ctypes as C
import re
import warnings

import numpy as np

from . import capi


def whitening_from_header(path, spreading_factor, low_data_rate):
    """The sequence the reference's block would select (lib/decode_impl.cc:72-113), read from a copy of its table header
    (include/lora/lora.h) that the USER supplies: `const unsigned short whitening_sequence_sf<SF>_[ldr_]implicit[N] = {...};`.
    The tables themselves are reference source and do not ship with this package."""
    name = f"whitening_sequence_sf{int(spreading_factor)}_{'ldr_' if low_data_rate else ''}implicit"
    with open(path) as f:
        for line in f:
            m = re.match(r"\s*const unsigned short " + name + r"\[\d+\]\s*=\s*\{(.*)\};", line)
            if m:
                return np.array([int(t, 0) for t in m.group(1).split(",") if t.strip()], np.uint16)
    raise KeyError(f"{name} not found in {path}")


class decode:
    name = "decode"                      # gr::block("decode", ...), lib/decode_impl.cc:52
    in_port, out_port = "in", "out"      # :66-67

    def __init__(self, spreading_factor=8, code_rate=4, low_data_rate=False, header=False, whitening_sequence=None):
        if not (5 < spreading_factor < 13):                     # asserts at :62-64
            raise ValueError("spreading_factor must be in 6..12")
        if not (0 < code_rate < 5):
            raise ValueError("code_rate must be in 1..4")
        self.spreading_factor, self.code_rate = int(spreading_factor), int(code_rate)
        self.low_data_rate, self.header = bool(low_data_rate), bool(header)   # header: accepted and ignored, as :115-119
        # DIVERGENCE from the reference, stated in INTEGRATION.md: lora.decode(sf, cr, ldr, header) always de-whitens with the
        # table it selects from sf and low_data_rate (:72-113, :155-162); those tables are not in this package, so the
        # sequence is an argument.  None (the default) de-whitens nothing and warns; False says "none" on purpose.
        if whitening_sequence is None:
            warnings.warn("decode(): no whitening_sequence given -- the reference block would de-whiten with its table for this "
                          "sf/low_data_rate, this one will not (pass the sequence, e.g. whitening_from_header(<your lora.h>, sf, "
                          "ldr), or whitening_sequence=False to silence this)", stacklevel=2)
        no_seq = whitening_sequence is None or whitening_sequence is False
        self._w = None if no_seq else np.ascontiguousarray(whitening_sequence, np.uint16)
        self._handlers = []

    @staticmethod
    def make(spreading_factor=8, code_rate=4, low_data_rate=False, header=False, whitening_sequence=None):
        return decode(spreading_factor, code_rate, low_data_rate, header, whitening_sequence)

    def set_msg_handler(self, fn):
        """fn(pdu) for every decoded PDU, pdu = ({}, numpy uint8 array): message_port_pub("out", ...), :511-512."""
        self._handlers.append(fn)

    def work(self, symbols):
        """decode_impl::decode (:420-513) for one symbol PDU (a uint16 sequence, or the ({}, array) pair)."""
        if isinstance(symbols, tuple):
            symbols = symbols[1]                                 # pmt::cdr(msg), :422
        s = np.ascontiguousarray(symbols, np.uint16)
        L = capi.lib()
        out = np.empty(L.lora_decode_max_bytes(len(s)), np.uint8)
        n = C.c_size_t(0)
        w = self._w
        rc = L.lora_decode_pdu(self.spreading_factor, self.code_rate, int(self.low_data_rate),
                               None if w is None else w.ctypes.data, 0 if w is None else len(w),
                               s.ctypes.data if len(s) else None, len(s), out.ctypes.data, len(out), C.byref(n))
        if rc:
            raise ValueError(f"lora_decode_pdu failed with code {rc}")
        res = out[:n.value].copy()
        for fn in self._handlers:
            fn(({}, res))
        return res
