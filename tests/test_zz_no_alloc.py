"""GroupNorm through the entry point that allocates nothing (its scratch is a fixed device array inside the
library) against the workspace-taking entry point: bit-identical outputs, two calls in a row, the second smaller."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(lib, check, x, gamma, beta, G, eps, silu, warm=False):
    B, C = x.shape[0], x.shape[-1]
    HW = x.numel() // (B * C)
    stream = torch.cuda.current_stream().cuda_stream
    nfl = lib.lbmk_groupnorm_scratch_floats(B, HW, C, G)
    assert 0 < nfl <= lib.lbmk_groupnorm_static_max_floats()
    ws = torch.empty(nfl, dtype=torch.float32, device=x.device)
    y_ws = torch.empty_like(x)
    check(lib.lbmk_groupnorm(x.data_ptr(), y_ws.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ws.data_ptr(),
                             B, HW, C, G, eps, silu, stream), "lbmk_groupnorm")
    y_st = torch.full_like(x, float("nan"))
    if warm:
        # the first call on a device resolves the array's address (and may load the code object): not part of the claim
        check(lib.lbmk_groupnorm_static(x.data_ptr(), y_st.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        B, HW, C, G, eps, silu, stream), "lbmk_groupnorm_static")
        y_st.fill_(float("nan"))
    torch.cuda.synchronize()
    # free device memory as the driver sees it (hipMemGetInfo): a hipMalloc inside the library would show here, which
    # torch.cuda.memory_allocated (torch's own caching allocator only) cannot see
    before_free = torch.cuda.mem_get_info()[0]
    before_torch = torch.cuda.memory_allocated()
    check(lib.lbmk_groupnorm_static(x.data_ptr(), y_st.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                    B, HW, C, G, eps, silu, stream), "lbmk_groupnorm_static")
    torch.cuda.synchronize()
    assert torch.cuda.mem_get_info()[0] == before_free
    assert torch.cuda.memory_allocated() == before_torch
    return y_ws, y_st


def test_groupnorm_static_matches_workspace_entry_point():
    from lbm_amd._C import LbmkError, check, lib
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(7)
    # first call: large; second call: smaller (the array keeps the first call's contents)
    for k, (B, HW, C, silu) in enumerate([(4, 64 * 64, 512, 1), (2, 20 * 12, 320, 0), (1, 8, 64, 1)]):
        x = (torch.randn(B, HW, C, generator=gen) * 2 + 0.5).to(torch.bfloat16).to(dev)
        g = torch.randn(C, generator=gen).to(dev)
        b = torch.randn(C, generator=gen).to(dev)
        y_ws, y_st = _both(lib, check, x, g, b, 32, 1e-5, silu, warm=k == 0)
        assert torch.equal(y_ws, y_st)
        ref = torch.nn.functional.group_norm(x.float().permute(0, 2, 1), 32, g, b, 1e-5)
        if silu:
            ref = torch.nn.functional.silu(ref)
        err = (y_st.float() - ref.permute(0, 2, 1)).norm() / ref.norm()
        assert err < 4e-3
    # on a stream that is not the default one (calls on one GPU share the array: one at a time, on whichever stream)
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        y_ws, y_st = _both(lib, check, x, g, b, 32, 1e-5, 1)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(y_ws, y_st)
    # beyond the built-in buffer: refused with a clear error, nothing launched
    B, HW, C = 8192, 16, 64
    assert lib.lbmk_groupnorm_scratch_floats(B, HW, C, 32) > lib.lbmk_groupnorm_static_max_floats()
    x = torch.zeros(B, HW, C, dtype=torch.bfloat16, device=dev)
    y = torch.empty_like(x)
    g = torch.ones(C, device=dev)
    with pytest.raises(LbmkError, match="scratch floats"):
        check(lib.lbmk_groupnorm_static(x.data_ptr(), y.data_ptr(), g.data_ptr(), g.data_ptr(), B, HW, C, 32, 1e-5, 0,
                                        torch.cuda.current_stream().cuda_stream), "lbmk_groupnorm_static")
