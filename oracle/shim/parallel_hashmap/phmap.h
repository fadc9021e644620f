// Stand-in for the un-vendored third_party/parallel-hashmap submodule of the
// reference (.gitmodules:1-3; directory is empty).  csrc/cpu/utils.h:4 includes
// parallel_hashmap/phmap.h in every CPU translation unit, but the hot-path TUs
// (spmm_cpu.cpp, convert_cpu.cpp) never instantiate a phmap container, so
// std::unordered_* aliases are enough to make the header parse.
// TEST INFRASTRUCTURE ONLY (used by oracle/build_ref.py).
#pragma once
#include <functional>
#include <unordered_map>
#include <unordered_set>
#include <utility>

namespace phmap {
template <class T> struct Hash : std::hash<T> {};
template <class A, class B> struct Hash<std::pair<A, B>> {
  size_t operator()(const std::pair<A, B> &p) const {
    return std::hash<A>()(p.first) * 1000003u ^ std::hash<B>()(p.second);
  }
};
template <class K, class V, class H = Hash<K>> using flat_hash_map = std::unordered_map<K, V, H>;
template <class K, class H = Hash<K>> using flat_hash_set = std::unordered_set<K, H>;
}  // namespace phmap
