// cu_budget.hpp — how many compute units the persistent kernels of this library size their grids for.
// Every hot kernel here is one (or two) persistent workgroup(s) per CU with most of the CU's LDS (gemm_nt8 160 KB, gemm_tn8 128 KB,
// the attention kernels 161 KB): a collective kernel (RCCL) launched beside them finds no CU to run on until one of those grids
// drains.  With world > 1 the data-parallel driver therefore RESERVES a few CUs (cream_cu_reserve, cream_amd/comm.py) and tells RCCL
// to use at most that many channels; every grid below is sized for the rest.
#pragma once
namespace cream {
int cu_count();          // physical CUs of the current device minus the reserve (a multiple of 8: one share per XCD), >= 8
}
