// Deferred auxiliary-stream launches (csrc/abi.hip): an entry point that ends in a reduction nobody on its stream consumes hands
// the launch over with swin_aux_push() while a block backward is collecting them.
#pragma once
#include <functional>

bool swin_aux_push(std::function<int(void*)> launch);
