"""hmd_nlist_build_head_list_noalloc (static device scratch, no allocation) against hmd_nlist_build_head_list
(caller-provided workspace), through the C ABI: two calls in a row, the second on a smaller input, plus the
size limit.  Bit-exact: the head list is integer work."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _heads(fn_name, pos, N, Nmax_t, with_workspace):
    from hoomd_amd import _abi
    from hoomd_amd._abi import check, ptr
    dev = pos.device
    head = torch.full((max(N, 1),), -1, dtype=torch.int64, device=dev)
    req = torch.full((1,), -1, dtype=torch.int64, device=dev)
    a = _abi.HeadListArgs()
    a.d_head_list, a.d_req_size_nlist, a.d_Nmax, a.d_pos = ptr(head), ptr(req), ptr(Nmax_t), ptr(pos)
    a.N, a.ntypes = N, Nmax_t.numel()
    ws = None
    if with_workspace:
        ws_bytes = _abi.lib.hmd_nlist_head_list_workspace_bytes(N)
        ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
        a.d_workspace, a.workspace_bytes = ptr(ws), ws_bytes
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    check(getattr(_abi.lib, fn_name)(C.byref(a), stream), fn_name)
    torch.cuda.synchronize()
    return head.cpu().numpy(), int(req.item())


def _case(N, ntypes, seed):
    from hoomd_amd import kernels as K
    rng = np.random.default_rng(seed)
    typeid = rng.integers(0, ntypes, N)
    pos = K.pack_pos(rng.uniform(-5, 5, (N, 3)), typeid)
    Nmax = rng.integers(1, 40, ntypes) * 4
    Nmax_t = torch.as_tensor(Nmax.astype(np.int32), device="cuda")
    return pos, typeid, Nmax, Nmax_t


@pytest.mark.parametrize("ntypes", [1, 3])
def test_noalloc_head_list_matches_workspace_entry_point_twice(ntypes):
    # first call: 300,001 particles = 147 tiles of the scan; second call: fewer tiles than the first left behind
    for N, seed in ((300001, 5), (4099, 6), (1, 7)):
        pos, typeid, Nmax, Nmax_t = _case(N, ntypes, seed)
        head_ref, req_ref = _heads("hmd_nlist_build_head_list", pos, N, Nmax_t, True)
        head, req = _heads("hmd_nlist_build_head_list_noalloc", pos, N, Nmax_t, False)
        assert req == req_ref
        assert np.array_equal(head, head_ref)
        # and against the definition (hoomd/md/NeighborList.cc:1544-1570)
        expect = np.concatenate([[0], np.cumsum(Nmax[typeid])])
        assert req == expect[-1]
        assert np.array_equal(head, expect[:-1])


def test_noalloc_head_list_rejects_sizes_beyond_its_scratch():
    from hoomd_amd import _abi
    pos, typeid, Nmax, Nmax_t = _case(16, 1, 1)
    a = _abi.HeadListArgs()
    head = torch.zeros(16, dtype=torch.int64, device="cuda")
    req = torch.zeros(1, dtype=torch.int64, device="cuda")
    a.d_head_list, a.d_req_size_nlist, a.d_Nmax, a.d_pos = _abi.ptr(head), _abi.ptr(req), _abi.ptr(Nmax_t), _abi.ptr(pos)
    a.N, a.ntypes = (1 << 24) + 1, 1
    assert _abi.lib.hmd_nlist_build_head_list_noalloc(C.byref(a), None) == _abi.ERR_TOO_LARGE
    assert b"16,777,216" in _abi.lib.hmd_last_error()


def test_noalloc_head_list_allocates_no_device_memory():
    """hipMemGetInfo before and after: once the first call on the GPU has happened (it loads the kernels and looks the
    array's address up), further calls leave the device's free memory where it was."""
    from hoomd_amd import _abi
    from hoomd_amd._abi import check, ptr
    N = 200003
    pos, typeid, Nmax, Nmax_t = _case(N, 2, 9)
    head = torch.zeros(N, dtype=torch.int64, device="cuda")
    req = torch.zeros(1, dtype=torch.int64, device="cuda")
    a = _abi.HeadListArgs()
    a.d_head_list, a.d_req_size_nlist, a.d_Nmax, a.d_pos = ptr(head), ptr(req), ptr(Nmax_t), ptr(pos)
    a.N, a.ntypes = N, 2
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(_abi.lib.hmd_nlist_build_head_list_noalloc(C.byref(a), stream), "warm call")
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for n in (N, 5000, N):
        a.N = n
        check(_abi.lib.hmd_nlist_build_head_list_noalloc(C.byref(a), stream), "hmd_nlist_build_head_list_noalloc")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == free0
    expect = np.concatenate([[0], np.cumsum(Nmax[typeid])])
    assert int(req.item()) == expect[-1] and np.array_equal(head.cpu().numpy(), expect[:-1])
