// TEMPORARY stubs until model.hip lands.
#include "common.h"
#include "../../include/mi355_retrieval.h"
using namespace mi355;
#define STUB(...) { set_error("model API not built yet"); return ERR_UNSUPPORTED; }
extern "C" {
int mi355_model_create(const char*, int, mi355_model_t*) STUB()
void mi355_model_destroy(mi355_model_t) {}
int mi355_model_num_tensors(mi355_model_t) STUB()
int mi355_model_tensor_info(mi355_model_t, int, const char**, int*, int64_t*, int*) STUB()
int mi355_model_feature_dim(mi355_model_t) STUB()
int mi355_model_num_classes(mi355_model_t) STUB()
int mi355_model_set_tensor(mi355_model_t, const char*, const float*, int64_t) STUB()
int mi355_model_pack(mi355_model_t, void*) STUB()
int mi355_model_forward_features(mi355_model_t, const float*, int, int, int, float*, float*, void*) STUB()
int mi355_model_forward(mi355_model_t, const float*, int, int, int, float*, float*, void*) STUB()
int mi355_model_enable_taps(mi355_model_t, int) STUB()
int mi355_model_read_tap(mi355_model_t, const char*, float*, int64_t, int64_t*, void*) STUB()
int mi355_model_traffic(mi355_model_t, int, int, int, double*, double*, double*) STUB()
int mi355_conv_input_silu(const float*, const float*, int, int, int, float*, void*) STUB()
}
