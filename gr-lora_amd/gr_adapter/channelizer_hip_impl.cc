// See channelizer_hip_impl.h.  The handle keeps the taps_per_filter - 1 samples of look-behind, the filter arm, the
// interpolation weight and the rotation phase itself, so GNU Radio's history mechanism is not used (history() stays
// 1) and a call consumes exactly the inputs whose outputs fit the output buffer.
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <gnuradio/io_signature.h>
#include "channelizer_hip_impl.h"

namespace gr {
namespace lora {

channelizer_hip_impl::sptr channelizer_hip_impl::make(double phase_inc, float rrate, const std::vector<float> &taps,
                                                      unsigned nfilts, float atten) {
  return gnuradio::get_initial_sptr(new channelizer_hip_impl(phase_inc, rrate, taps, nfilts, atten));
}

channelizer_hip_impl::channelizer_hip_impl(double phase_inc, float rrate, const std::vector<float> &taps,
                                           unsigned nfilts, float atten)
    : gr::block("channelizer_hip", gr::io_signature::make(1, 1, sizeof(gr_complex)),
                gr::io_signature::make(1, 1, sizeof(gr_complex))),
      d_chan(NULL), d_rate(rrate) {
  std::vector<float> t(taps);
  if (t.empty()) {                       // the GRC field left empty: pfb.arb_resampler_ccf.create_taps
    size_t n = 0;
    int rc = lora_hip_chan_design_taps(rrate, nfilts, atten, NULL, 0, &n);
    if (rc == LORA_HIP_OK) { t.resize(n); rc = lora_hip_chan_design_taps(rrate, nfilts, atten, &t[0], n, &n); }
    if (rc != LORA_HIP_OK) throw std::runtime_error(std::string("lora_hip_chan_design_taps: ") + lora_hip_strerror(rc));
  }
  const char *dev = getenv("LORA_HIP_DEVICE");
  const int rc = lora_hip_chan_create(&d_chan, dev ? atoi(dev) : 0, phase_inc, rrate, &t[0], t.size(), nfilts);
  if (rc != LORA_HIP_OK) throw std::runtime_error(std::string("lora_hip_chan_create: ") + lora_hip_strerror(rc));
  set_relative_rate(rrate);              // as pfb_arb_resampler_ccf_impl does
}

channelizer_hip_impl::~channelizer_hip_impl() { lora_hip_chan_destroy(d_chan); }

void channelizer_hip_impl::forecast(int noutput_items, gr_vector_int &ninput_items_required) {
  ninput_items_required[0] = (int)std::ceil(noutput_items / d_rate) + 1;
}

int channelizer_hip_impl::general_work(int noutput_items, gr_vector_int &ninput_items,
                                       gr_vector_const_void_star &input_items, gr_vector_void_star &output_items) {
  // the most inputs whose outputs fit (the output count is known before any work is done: the index recurrence of
  // pfb_arb_resampler_ccf::filter is data-independent); bisection on a monotone count
  size_t lo = 0, hi = (size_t)ninput_items[0];
  if (lora_hip_chan_output_count(d_chan, hi) > (size_t)noutput_items) {
    while (hi - lo > 1) {
      const size_t mid = lo + (hi - lo) / 2;
      if (lora_hip_chan_output_count(d_chan, mid) <= (size_t)noutput_items) lo = mid; else hi = mid;
    }
  } else lo = hi;
  size_t produced = 0;
  const int rc = lora_hip_chan_work(d_chan, static_cast<const lora_c32 *>(input_items[0]), lo,
                                    static_cast<lora_c32 *>(output_items[0]), (size_t)noutput_items, &produced, 0);
  if (rc != LORA_HIP_OK) {
    // nothing is consumed and the flowgraph stops: dropping the input would hand the demod a stream with a hole in it
    std::cerr << "lora_hip_chan_work: " << lora_hip_strerror(rc) << " " << lora_hip_chan_last_hip_error(d_chan) << std::endl;
    return WORK_DONE;
  }
  if (lo == 0 && ninput_items[0] > 0) {
    // not even one input's outputs fit (rate > 1): ask the scheduler for more output space instead of spinning
    set_output_multiple((int)std::ceil(d_rate) + 1);
    return 0;
  }
  consume_each((int)lo);
  return (int)produced;
}

}  // namespace lora
}  // namespace gr
