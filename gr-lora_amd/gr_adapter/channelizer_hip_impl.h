// GNU Radio adapter: one block in place of blocks.rotator_cc + pfb.arb_resampler_ccf in front of lora.demod
// (examples/rx_usrp.grc:489-503, :825-860), implemented on lora_hip_chan (include/lora_hip_chan.h).
// Built only where GNU Radio 3.7 is installed (not in the development image).  A flowgraph replaces
//   blocks.rotator_cc(phase_inc) -> pfb.arb_resampler_ccf(rrate, taps=None, flt_size=32)       with
//   lora.channelizer_hip(phase_inc, rrate, taps, nfilts, atten)
// (a new block: the two it stands for belong to GNU Radio, not to gr-lora, so nothing of gr-lora is replaced).
#ifndef INCLUDED_LORA_CHANNELIZER_HIP_IMPL_H
#define INCLUDED_LORA_CHANNELIZER_HIP_IMPL_H

#include <vector>
#include <gnuradio/block.h>
#include "lora_hip_chan.h"

namespace gr {
namespace lora {

class channelizer_hip_impl : public gr::block {
 public:
  typedef boost::shared_ptr<channelizer_hip_impl> sptr;
  // taps empty: designed as pfb.arb_resampler_ccf does for rate < 1 (firdes.low_pass_2, Blackman-harris)
  static sptr make(double phase_inc, float rrate, const std::vector<float> &taps, unsigned nfilts = 32,
                   float atten = 100.0f);
  channelizer_hip_impl(double phase_inc, float rrate, const std::vector<float> &taps, unsigned nfilts, float atten);
  ~channelizer_hip_impl();

  void forecast(int noutput_items, gr_vector_int &ninput_items_required);
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items);

 private:
  lora_hip_chan_t *d_chan;
  float d_rate;
};

}  // namespace lora
}  // namespace gr
#endif
