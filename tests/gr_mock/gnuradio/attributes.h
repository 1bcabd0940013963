// tests/gr_mock: a stand-in for exactly the GNU Radio 3.7 surface gr-lora_amd/gr_adapter/*.cc use, for `g++ -fsyntax-only`
// (tests/test_gr_adapter_syntax.py).  Declarations only: a type check of the adapters, nothing of GNU Radio's behaviour.
#ifndef MOCK_GNURADIO_ATTRIBUTES_H
#define MOCK_GNURADIO_ATTRIBUTES_H
#define __GR_ATTR_EXPORT __attribute__((visibility("default")))
#define __GR_ATTR_IMPORT __attribute__((visibility("default")))
#endif
