// Compile-only check (tests/test_abi.py::test_rccl_declarations_match_the_installed_header): the engine binds librccl at
// run time through function pointers it declares by hand (suitesparse_amd/csrc/multi.hip.hpp: enum values as int, the
// communicator as void *, the unique id as 128 bytes by value).  This file states the same facts against the installed
// rccl.h, so a header that moves away from them fails here and not on the first multi-GPU node.
// F64 / I64 / SUM / MIN come from the engine's source (-D on the command line).
#include <rccl/rccl.h>
#include <type_traits>
static_assert(sizeof(ncclUniqueId) == 128, "id");
static_assert((int)ncclFloat64 == F64 && (int)ncclInt64 == I64 && (int)ncclSum == SUM && (int)ncclMin == MIN, "enums");
static_assert(sizeof(ncclDataType_t) == sizeof(int) && sizeof(ncclRedOp_t) == sizeof(int) && sizeof(ncclResult_t) == sizeof(int), "enum size");
static_assert(std::is_pointer<ncclComm_t>::value, "comm");
static_assert(std::is_same<decltype(&ncclGetUniqueId), ncclResult_t (*)(ncclUniqueId *)>::value, "GetUniqueId");
static_assert(std::is_same<decltype(&ncclCommInitRank), ncclResult_t (*)(ncclComm_t *, int, ncclUniqueId, int)>::value, "CommInitRank");
static_assert(std::is_same<decltype(&ncclCommDestroy), ncclResult_t (*)(ncclComm_t)>::value, "CommDestroy");
static_assert(std::is_same<decltype(&ncclCommSplit), ncclResult_t (*)(ncclComm_t, int, int, ncclComm_t *, ncclConfig_t *)>::value, "CommSplit");
static_assert(std::is_same<decltype(&ncclReduce), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t)>::value, "Reduce");
static_assert(std::is_same<decltype(&ncclBroadcast), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t)>::value, "Broadcast");
static_assert(std::is_same<decltype(&ncclAllReduce), ncclResult_t (*)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)>::value, "AllReduce");
static_assert(std::is_same<decltype(&ncclGroupStart), ncclResult_t (*)()>::value, "GroupStart");
static_assert(std::is_same<decltype(&ncclGroupEnd), ncclResult_t (*)()>::value, "GroupEnd");
static_assert(std::is_same<decltype(&ncclGetErrorString), const char *(*)(ncclResult_t)>::value, "GetErrorString");
int main() { return 0; }
