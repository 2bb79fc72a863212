// TEMPORARY stubs until ssd_conv.hip / ssd_net.hip land (removed in the next commits).
#include "common.h"
#define U(name) ssd::set_error(#name ": not built yet"); return SSD_E_UNSUPPORTED
extern "C" {
int ssd_same_pads(int, int, int, int, int*, int*) { U(ssd_same_pads); }
int ssd_conv_out_size(int, int, int, int, int, int) { U(x); }
size_t ssd_conv_packed_weight_floats(int, int, int, int) { return 0; }
int ssd_conv_pack_weights(const float*, int, int, int, int, float*, void*) { U(x); }
int ssd_conv2d(const ssd_conv_desc*, const float*, const float*, const float*, const float*, const float*, float*, long, long, void*) { U(x); }
int ssd_dwconv3x3(const float*, int, int, int, int, int, int, int, int, int, const float*, const float*, const float*, int, float*, void*) { U(x); }
int ssd_maxpool2d(const float*, int, int, int, int, int, int, int, int, int, int, float*, void*) { U(x); }
int ssd_l2norm(const float*, long, int, const float*, float*, void*) { U(x); }
int ssd_softmax(const float*, long, int, float*, void*) { U(x); }
ssd_net* ssd_net_create(int, int, int, const int*, int) { return nullptr; }
void ssd_net_destroy(ssd_net*) {}
int ssd_net_num_params(const ssd_net*) { return 0; }
const char* ssd_net_param_name(const ssd_net*, int) { return ""; }
int ssd_net_param_rank(const ssd_net*, int) { return 0; }
const int* ssd_net_param_shape(const ssd_net*, int) { return nullptr; }
int ssd_net_set_param(ssd_net*, const char*, const float*, size_t) { U(x); }
int ssd_net_get_param(const ssd_net*, const char*, float*, size_t) { U(x); }
int ssd_net_finalize(ssd_net*, int) { U(x); }
int ssd_net_num_priors(const ssd_net*) { return 0; }
int ssd_net_feature_map_size(const ssd_net*, int) { return 0; }
int ssd_net_forward(ssd_net*, const float*, int, float*, float*, void*) { U(x); }
int ssd_net_predict(ssd_net*, const float*, int, const float*, const float*, int, float, float, float*, float*, float*, int*, void*) { U(x); }
long ssd_net_fetch_activation(ssd_net*, const char*, float*, size_t) { return -3; }
int ssd_net_num_layers(const ssd_net*) { return 0; }
const char* ssd_net_layer_name(const ssd_net*, int) { return ""; }
const char* ssd_net_layer_kind(const ssd_net*, int) { return ""; }
double ssd_net_layer_flops(const ssd_net*, int, int) { return 0; }
double ssd_net_layer_bytes(const ssd_net*, int, int) { return 0; }
int ssd_net_profile_layers(ssd_net*, const float*, int, int, float*, void*) { U(x); }
}
