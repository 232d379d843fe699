// Internal interface between the two chain interpreters: npf_chain_run (chain_kernel.hip) hands a validated
// bf16 program (prog->reserved[2] == 1) to the bf16 interpreter of chain16_kernel.hip.
#pragma once
#include <type_traits>

#include "npf_common.hpp"

int npf16_chain_launch(const npf_program_t& g, void* stream);
