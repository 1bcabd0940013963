"""Raw capture files on either side of the demod path.

The reference's flowgraphs read and write captures with GNU Radio's file source/sink blocks as raw
interleaved little-endian float32 I/Q, no header (examples/txrx_sim.grc:219-269, "*.32iq"); that is
exactly numpy complex64 on disk.  demod_file() streams such a file through the block in bounded
chunks, so a capture larger than host memory works; chunking does not change the PDUs
(lora_demod_core buffers what a call cannot consume yet).
"""
import numpy as np


def write_iq(path, samples):
    np.ascontiguousarray(samples, np.complex64).tofile(path)


def read_iq(path, mmap=True):
    """complex64 view of a raw float32 I/Q file (memory-mapped by default)."""
    if mmap:
        return np.memmap(path, dtype=np.complex64, mode="r")
    return np.fromfile(path, dtype=np.complex64)


def iter_chunks(path, chunk_samples=1 << 22):
    x = read_iq(path)
    for k in range(0, len(x), chunk_samples):
        yield np.asarray(x[k:k + chunk_samples])


def demod_file(path, block, chunk_samples=1 << 22, flush_symbols=4, channelizer=None):
    """Feed a capture file to `block` (gr-lora_amd.demod or anything with work()); returns the PDUs.
    flush_symbols*N zeros are appended so that a packet ending at the end of the file is flushed (GNU
    Radio's scheduler only calls the block while 4N-1 samples are available; a live source keeps going).
    With `channelizer` (gr-lora_amd.channelizer) the file is a wideband capture: every chunk goes to the GPU once,
    is shifted and resampled there, and the block reads the result in HBM (examples/rx_usrp.grc's chain from a
    file instead of a USRP)."""
    pdus = []
    if channelizer is None:
        for chunk in iter_chunks(path, chunk_samples):
            pdus += block.work(chunk)
    else:
        import torch
        dev = torch.device("cuda", getattr(channelizer, "device", 0))
        for chunk in iter_chunks(path, chunk_samples):
            pdus += block.work(channelizer.work(torch.tensor(chunk, device=dev)))    # one host-to-device copy per chunk
    if flush_symbols:
        n = 1 << block.spreading_factor
        pdus += block.work(np.zeros(flush_symbols * n, np.complex64))
    return pdus
