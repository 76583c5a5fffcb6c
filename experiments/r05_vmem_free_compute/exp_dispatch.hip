// Experiment build only: fr_conv3x3_strip of libfrhip_exp.so.  The product's entry point is compiled under the name
// fr_conv3x3_strip_product (the Makefile here renames it on the command line, no product source is touched); this one
// hands 256 -> 256 @14x14, B > 160 to the experiment kernels when their switch is on and everything else to the product.
//   FRHIP_RELAY=1  conv3x3_relay.hip      FRHIP_SOLO=1  conv3x3_solo.hip      (both 0: the product's strip kernel)
#include "common.h"
#include "frhip_internal.h"

extern "C" int fr_conv3x3_strip_product(const FrConvArgs* a, void* stream);
bool fr_solo_serves(const FrConvArgs& a);
int fr_solo_launch(const FrConvArgs& a, hipStream_t st);
bool fr_relay_serves(const FrConvArgs& a);
int fr_relay_launch(const FrConvArgs& a, hipStream_t st);

extern "C" int fr_conv3x3_strip(const FrConvArgs* a, void* stream) {
  if (a && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1) {
    if (fr_relay_serves(*a)) return fr_relay_launch(*a, (hipStream_t)stream);
    if (fr_solo_serves(*a)) return fr_solo_launch(*a, (hipStream_t)stream);
  }
  return fr_conv3x3_strip_product(a, stream);
}
