// Builds against include/xdtts_host.hpp + libxdtts_hip.so with plain g++ (no HIP headers): proves the
// boundary is a C ABI.  Prints the reference's char-id known answer (src/tacotron2/mod.rs:494-508);
// with a GPU present also runs Tacotron2::infer(&[Unit::Character('a')]) like mod.rs:511-522.
#include <cstdio>
#include <string>

#include "xdtts_host.hpp"

int main() {
  const std::string text = "hello world!";
  for (char c : text) {
    const std::string tok(1, c);
    std::printf("%lld ", (long long)xdtts_unit_id(tok.c_str(), c == '!' ? 0 : 1));
  }
  std::printf("\n");
  if (xdtts_device_count() > 0) {
    xdtts::Tacotron2 model = xdtts::Tacotron2::synthetic();
    xdtts_infer_opts o;
    xdtts_infer_opts_default(&o);
    o.max_steps = 5;
    xdtts::Array2 spec = model.infer({{"a", true}}, &o);
    std::printf("spec %zu x %zu\n", spec.rows, spec.cols);
    xdtts::GriffinLim voc = xdtts::create_griffin_lim();
    std::printf("audio %zu\n", voc.infer(spec).size());
    // XdTts::infer for two texts in one call (xdtts_synthesize_batch)
    const std::vector<std::vector<xdtts::Unit>> texts = {{{"a", true}, {"b", true}}, {{"HH", false}, {"AH0", false}, {" ", false}, {"L", false}}};
    const auto many = xdtts::infer_many(model, voc, texts, &o);
    std::printf("many %zu:", many.size());
    for (const auto &m : many) std::printf(" %zu/%zu", m.first.cols, m.second.size());
    std::printf("\n");
  } else {
    try {
      xdtts::Tacotron2::synthetic();
      std::printf("unexpected success\n");
      return 1;
    } catch (const xdtts::Error &e) {
      std::printf("no device: status %d\n", (int)e.status);
    }
  }
  return 0;
}
