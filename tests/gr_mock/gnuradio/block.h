// tests/gr_mock: see attributes.h.  gr::block, pmt and boost::shared_ptr as far as the adapters touch them.
#ifndef MOCK_GNURADIO_BLOCK_H
#define MOCK_GNURADIO_BLOCK_H
#include <complex>
#include <cstddef>
#include <memory>
#include <string>
#include <vector>
#include <stdint.h>

namespace boost { template <typename T> using shared_ptr = std::shared_ptr<T>; }
typedef std::complex<float> gr_complex;
typedef std::vector<int> gr_vector_int;
typedef std::vector<const void *> gr_vector_const_void_star;
typedef std::vector<void *> gr_vector_void_star;

namespace pmt {
struct pmt_base;
typedef boost::shared_ptr<pmt_base> pmt_t;
pmt_t mp(const std::string &s);
pmt_t cons(const pmt_t &x, const pmt_t &y);
pmt_t make_dict();
pmt_t init_u16vector(size_t k, const uint16_t *data);
}  // namespace pmt

namespace gr {
class io_signature {
 public:
  typedef boost::shared_ptr<io_signature> sptr;
  static sptr make(int min_streams, int max_streams, int sizeof_stream_item);
};
class block {
 public:
  enum { WORK_CALLED_PRODUCE = -2, WORK_DONE = -1 };
  virtual ~block();
  virtual void forecast(int noutput_items, gr_vector_int &ninput_items_required);
  virtual int general_work(int noutput_items, gr_vector_int &ninput_items, gr_vector_const_void_star &input_items,
                           gr_vector_void_star &output_items);
  void consume_each(int how_many_items);
  void set_relative_rate(double relative_rate);
  void set_output_multiple(int multiple);
  void message_port_register_out(pmt::pmt_t port_id);
  void message_port_pub(pmt::pmt_t port_id, pmt::pmt_t msg);
 protected:
  block() {}
  block(const std::string &name, io_signature::sptr input_signature, io_signature::sptr output_signature);
};
}  // namespace gr

namespace gnuradio {
template <class T> boost::shared_ptr<T> get_initial_sptr(T *p) { return boost::shared_ptr<T>(p); }
}
#endif
