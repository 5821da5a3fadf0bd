// Stand-in for search_device.hip on a box without HIP: every device entry reports "no device" (what the product library does
// there, too) -- so that the host side of the C ABI (capi.cpp, builder.cpp, hnswio.cpp, datamap.cpp) can be built with
// AddressSanitizer and driven by the CPU test-suite.  Test infrastructure only.
#include "search_device.hpp"
#include "hnswio.hpp"
namespace hnswgpu {
struct DeviceIndex::Workspace {};
static const char* kNoDev = "no HIP device visible (a gfx950 GPU is required; there is no CPU fallback)";
DeviceIndex::DeviceIndex() {}
DeviceIndex::~DeviceIndex() {}
int DeviceIndex::upload(const FlatIndex&, int, std::string& err) { err = kNoDev; return ERR_DEVICE; }
int DeviceIndex::search_device(const float*, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t*, float*, uint8_t*, int32_t*, uint32_t*, uint32_t*,
                               void*, const uint64_t*, uint64_t, CallInfo*, std::string& err, const RowFeed*, OutLayout) { err = kNoDev; return ERR_DEVICE; }
int DeviceIndex::search_host(const float*, uint64_t, uint64_t, uint64_t, uint64_t, uint64_t*, float*, uint8_t*, int32_t*, uint32_t*,
                             const uint64_t*, uint64_t, bool, uint8_t*, CallInfo*, std::string& err) { err = kNoDev; return ERR_DEVICE; }
int DeviceIndex::search_host_staged(const float*, const float* const*, uint64_t, uint64_t, uint64_t, uint64_t, const uint64_t*, uint64_t, bool,
                                    bool, const AnswerSink&, CallInfo*, std::string& err) { err = kNoDev; return ERR_DEVICE; }
int DeviceIndex::kernel_metric() const { return dist_; }
CallInfo DeviceIndex::last_call() const { return CallInfo{}; }
int device_count() { return 0; }
const Knobs& knobs() { static const Knobs k; return k; }  // (the hooks live in search_device.hip: defaults here)
void reload_knobs() {}
void* pinned_alloc(size_t, void**) { return nullptr; }  // (no device: the callers fall back to ordinary memory)
void pinned_free(void*) {}
namespace {
class NoDeviceBackend : public BuildSearchBackend {
public:
    int check(uint64_t, std::string& err) override { err = kNoDev; return ERR_DEVICE; }
    int begin(const float* const*, uint64_t, uint64_t, uint64_t, const uint8_t*, int, uint64_t, uint64_t, unsigned, uint64_t, std::string& err) override { err = kNoDev; return ERR_DEVICE; }
    uint32_t rec_words() const override { return 2; }
    uint32_t* patch_buffer(uint64_t, std::string& err) override { err = kNoDev; return nullptr; }
    int patch(uint64_t, std::string& err) override { err = kNoDev; return ERR_DEVICE; }
    int search_window(uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, const WindowSelect&, WindowSearchResults&, std::string& err) override { err = kNoDev; return ERR_DEVICE; }
};
}  // namespace
std::unique_ptr<BuildSearchBackend> make_device_build_backend(int) { return std::unique_ptr<BuildSearchBackend>(new NoDeviceBackend()); }
int eval_distance_matrix_device(int, const float*, uint64_t, const float*, uint64_t, uint64_t, uint32_t, bool, float*, std::string& err, int) { err = kNoDev; return ERR_DEVICE; }
int gather_sharded_answers(const int*, int, const uint64_t*, uint64_t, const uint64_t* const*, const float* const*, const uint8_t* const*, const int32_t* const*,
                           const uint32_t* const*, int, uint64_t*, float*, uint8_t*, int32_t*, uint32_t*, void*, std::string& err) { err = kNoDev; return ERR_DEVICE; }
int lane_lab_device(int, uint32_t, uint32_t, uint32_t, uint32_t, const uint32_t*, uint32_t, const uint32_t*, uint32_t, uint32_t*, uint32_t, std::string& err) { err = kNoDev; return ERR_DEVICE; }
}  // namespace hnswgpu
