// TEST STUB (tests/test_abi_cpu.py): the few at:: names INTEGRATION.md's binding snippet touches, so that the snippet's calls into
// include/valor_hip.h are type-checked by the compiler (-fsyntax-only) without the real torch headers.
#pragma once
#include <cstdint>
#include <initializer_list>
#include <vector>
namespace at {
enum ScalarType { kFloat, kBFloat16 };
struct TensorOptions { TensorOptions dtype(ScalarType) const { return *this; } };
struct Tensor {
    void* data_ptr() const { return nullptr; }
    template <typename T> T* data_ptr() const { return nullptr; }
    int64_t numel() const { return 0; }
    int64_t size(int) const { return 0; }
    ScalarType scalar_type() const { return kFloat; }
    TensorOptions options() const { return {}; }
};
inline Tensor empty_like(const Tensor&) { return {}; }
inline Tensor empty(std::initializer_list<int64_t>, TensorOptions) { return {}; }
}  // namespace at
struct StubModule { template <typename F> void def(const char*, F) {} };
#define TORCH_CHECK(cond, ...) do { if (!(cond)) { } } while (0)
#define TORCH_EXTENSION_NAME stub_ext
#define PYBIND11_MODULE(name, m) static void pybind_init_stub(StubModule& m)
