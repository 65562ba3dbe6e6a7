// errors.h — thread-local last-error text behind tsgo_last_error().
#pragma once
#include <string>
namespace tsgo {
int set_error(int code, const std::string& text);   // returns code
const char* last_error();
}  // namespace tsgo
