// GNU Radio adapter: gr::lora::demod implemented on lora_demod_core (MI355X hot path).
// Built only where GNU Radio 3.7 is installed (not in the development image); it replaces
// lib/demod_impl.{h,cc} of gr-lora and leaves include/lora/demod.h, the SWIG binding and the
// GRC block definition untouched, so flowgraphs keep calling
//   lora.demod(spreading_factor, low_data_rate, beta, fft_factor)      (grc/lora_demod.xml:7)
#ifndef INCLUDED_LORA_DEMOD_HIP_IMPL_H
#define INCLUDED_LORA_DEMOD_HIP_IMPL_H

#include <lora/demod.h>          // the reference's public header, unchanged
#include "lora_demod_core.h"

namespace gr {
namespace lora {

class demod_hip_impl : public demod {
 public:
  demod_hip_impl(unsigned short spreading_factor, bool low_data_rate, float beta, unsigned short fft_factor);
  ~demod_hip_impl();

  void forecast(int noutput_items, gr_vector_int &ninput_items_required);
  int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                   gr_vector_void_star &output_items);

 private:
  static void on_pdu(void *self, const uint16_t *symbols, size_t n);
  lora_demod_core_t *d_core;
  pmt::pmt_t d_out_port;
  unsigned short d_sf;
};

}  // namespace lora
}  // namespace gr
#endif
