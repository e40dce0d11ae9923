// hsw_nounwind.hpp -- the exception barrier of the C ABI (include/hsw.h: "nothing unwinds across the
// boundary").  Every int-returning entry point is a function-try-block closed by HSW_NO_UNWIND, so a
// failed host allocation (std::vector, std::string, new) comes back as a status code instead of
// propagating into a caller that cannot unwind (Rust FFI, C, ctypes).
#ifndef HSW_NOUNWIND_HPP
#define HSW_NOUNWIND_HPP

#include <new>
#include <stdexcept>

#define HSW_NO_UNWIND                                                   \
    catch (const std::bad_alloc &) { return HSW_ERR_NOMEM; }            \
    catch (const std::length_error &) { return HSW_ERR_TOO_LARGE; }     \
    catch (...) { return HSW_ERR_INVALID_ARG; }

#endif
