"""gr-lora_amd: MI355X-native hot path of gr-lora's demod block (dechirp, window, zero-padded
FFT, |X|^2 argmax) behind the reference's `demod` API.  The directory name has a hyphen (it
mirrors the reference's repository name), so import it through the `gr_lora_amd` shim at the
repository root or with importlib.import_module("gr-lora_amd")."""
from . import capi  # noqa: F401
from .capi import HipDemod, LoraHipError, make_chirp_tables, make_kaiser, make_desc  # noqa: F401
from .demod import demod  # noqa: F401,E402
from .channelizer import channelizer, create_taps  # noqa: F401,E402
from .decode import decode, whitening_from_header  # noqa: F401,E402
