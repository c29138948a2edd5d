// Minimal tl::expected / tl::unexpected work-alike for builds without the
// tl-expected package (value-or-error with has_value / value / error).
#pragma once
#if __has_include(<tl/expected.hpp>) && !defined(VO_FORCE_COMPAT_TYPES)
#include <tl/expected.hpp>
#else
#include <optional>
#include <stdexcept>
#include <utility>
#define VO_COMPAT_EXPECTED 1
namespace tl {

template <typename E>
class unexpected {
public:
    explicit unexpected(E e) : e_(std::move(e)) {}
    E const& value() const& { return e_; }
    E&& value() && { return std::move(e_); }

private:
    E e_;
};
template <typename E> unexpected(E) -> unexpected<E>;

template <typename T, typename E>
class expected {
public:
    expected(T v) : v_(std::move(v)) {}   // NOLINT: implicit, like tl::expected
    template <typename G> expected(unexpected<G> u) : e_(E(std::move(u).value())) {}   // NOLINT
    [[nodiscard]] bool has_value() const noexcept { return v_.has_value(); }
    explicit operator bool() const noexcept { return has_value(); }
    T& value() { if (!v_) throw std::logic_error("expected holds an error"); return *v_; }
    T const& value() const { if (!v_) throw std::logic_error("expected holds an error"); return *v_; }
    T& operator*() { return *v_; }
    T const& operator*() const { return *v_; }
    T* operator->() { return &*v_; }
    T const* operator->() const { return &*v_; }
    E const& error() const { return *e_; }

private:
    std::optional<T> v_;
    std::optional<E> e_;
};

// value-less flavour: success or an error
template <typename E>
class expected<void, E> {
public:
    expected() = default;
    template <typename G> expected(unexpected<G> u) : e_(E(std::move(u).value())) {}   // NOLINT
    [[nodiscard]] bool has_value() const noexcept { return !e_.has_value(); }
    explicit operator bool() const noexcept { return has_value(); }
    void value() const { if (e_) throw std::logic_error("expected holds an error"); }
    E const& error() const { return *e_; }

private:
    std::optional<E> e_;
};

}  // namespace tl
#endif
