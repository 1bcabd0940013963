// See demod_hip_impl.h.  The block keeps the reference's name, ports and PDU format
// (lib/demod_impl.cc:58-60, :74-75, :420-422) and hands every new input item to the core, which
// keeps its own 3N-1 samples of look-behind; GNU Radio's history mechanism is therefore not used
// (history() stays 1) and every call consumes all it was given.
#include <cstdlib>
#include <iostream>
#include <stdexcept>
#include <string>
#include <gnuradio/io_signature.h>
#include "demod_hip_impl.h"

namespace gr {
namespace lora {

demod::sptr demod::make(unsigned short spreading_factor, bool low_data_rate, float beta, unsigned short fft_factor) {
  return gnuradio::get_initial_sptr(new demod_hip_impl(spreading_factor, low_data_rate, beta, fft_factor));
}

demod_hip_impl::demod_hip_impl(unsigned short spreading_factor, bool low_data_rate, float beta,
                               unsigned short fft_factor)
    : gr::block("demod", gr::io_signature::make(1, 1, sizeof(gr_complex)), gr::io_signature::make(0, 0, 0)),
      d_core(NULL), d_sf(spreading_factor) {
  d_out_port = pmt::mp("out");
  message_port_register_out(d_out_port);
  const char *dev = getenv("LORA_HIP_DEVICE");
  const int rc = lora_demod_core_create(&d_core, dev ? atoi(dev) : 0, spreading_factor, low_data_rate, beta, fft_factor);
  if (rc != LORA_HIP_OK) throw std::runtime_error(std::string("lora_demod_core_create: ") + lora_hip_strerror(rc));
}

demod_hip_impl::~demod_hip_impl() { lora_demod_core_destroy(d_core); }

void demod_hip_impl::forecast(int noutput_items, gr_vector_int &ninput_items_required) {
  ninput_items_required[0] = noutput_items * (1 << d_sf);       // as lib/demod_impl.cc:142-147
}

void demod_hip_impl::on_pdu(void *self, const uint16_t *symbols, size_t n) {
  demod_hip_impl *me = static_cast<demod_hip_impl *>(self);
  pmt::pmt_t output = pmt::init_u16vector(n, symbols);
  me->message_port_pub(me->d_out_port, pmt::cons(pmt::make_dict(), output));
}

int demod_hip_impl::general_work(int noutput_items, gr_vector_int &ninput_items,
                                 gr_vector_const_void_star &input_items, gr_vector_void_star &) {
  const lora_c32 *in = static_cast<const lora_c32 *>(input_items[0]);
  const int rc = lora_demod_core_work(d_core, in, (size_t)ninput_items[0], &demod_hip_impl::on_pdu, this);
  if (rc != LORA_HIP_OK) {
    // a failed call may have emitted some PDUs already; the samples are not consumed and the flowgraph stops
    std::cerr << "lora_demod_core_work: " << lora_demod_core_last_error(d_core) << std::endl;
    return WORK_DONE;
  }
  consume_each(ninput_items[0]);
  return noutput_items;
}

}  // namespace lora
}  // namespace gr
