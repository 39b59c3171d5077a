// TEST STUB: see tests/stubs/torch/extension.h
#pragma once
namespace at { namespace hip {
struct Stream { void* stream() const { return nullptr; } };
inline Stream getCurrentHIPStream() { return {}; }
}}
