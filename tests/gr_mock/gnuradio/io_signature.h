// tests/gr_mock: see attributes.h.
#ifndef MOCK_GNURADIO_IO_SIGNATURE_H
#define MOCK_GNURADIO_IO_SIGNATURE_H
#include <gnuradio/block.h>
#endif
