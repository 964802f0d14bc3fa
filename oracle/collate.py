"""Oracle: CollateProducts.  TEST INFRASTRUCTURE ONLY.

Restates ``CollateProducts.process`` (reference ``draco/analysis/transform.py:168-330``) and the
helpers it leans on -- ``TelescopeStreamMixIn.setup`` (``:99-139``), ``tools.find_inputs``
(``util/tools.py:130-169``), ``tools.calculate_redundancy`` (``:313-352`` + the Cython loop
``_fast_tools.pyx:134-203``), ``tools.redefine_stack_index_map`` (``:359-414``) -- on plain arrays,
for unstacked inputs and for already redundancy-stacked ones (``transform.py:206-221``).
Pinned by ``tests/golden/transform_collate.npz`` (outputs of the reference class).
"""

from __future__ import annotations

import numpy as np

from .mapmaker import find_keys
from .transform import invert_no_zero


def cmap(i, j, n):
    if i > j:
        i, j = j, i
    return (n * (n + 1) // 2) - ((n - i) * (n - i + 1) // 2) + (j - i)


def telescope_maps(nfeed, uniquepairs, feedmap, feedconj, feedmask, npairs):
    """``bt_stack``, ``bt_prod``, ``bt_rev`` of ``TelescopeStreamMixIn.setup`` (``transform.py:110-139``)."""
    bt_stack = np.array(
        [(cmap(a, b, nfeed), 0) if a <= b else (cmap(b, a, nfeed), 1) for a, b in uniquepairs],
        dtype=[("prod", "<u4"), ("conjugate", "u1")],
    )
    triu = np.triu_indices(nfeed)
    bt_prod = np.zeros(len(triu[0]), dtype=[("input_a", "<u2"), ("input_b", "<u2")])
    bt_prod["input_a"], bt_prod["input_b"] = triu
    fm = feedmask[triu]
    bt_rev = np.empty(fm.size, dtype=[("stack", "<u4"), ("conjugate", "u1")])
    bt_rev["stack"] = np.where(fm, feedmap[triu], npairs)
    bt_rev["conjugate"] = np.where(fm, feedconj[triu], 0)
    return bt_prod, bt_stack, bt_rev


def calculate_redundancy(input_flags, prod, stack_index, nstack):
    """``redundancy[stack, t] = sum_{prod in stack} flag[a, t] * flag[b, t]`` (all flags taken good if none set)."""
    input_flags = np.asarray(input_flags, dtype=np.float32)
    if not np.any(input_flags):
        input_flags = np.ones_like(input_flags)
    red = np.zeros((nstack, input_flags.shape[1]), dtype=np.float32)
    for ii in range(len(prod)):
        ist = int(stack_index[ii])
        if 0 <= ist < nstack:
            red[ist] += input_flags[prod["input_a"][ii]] * input_flags[prod["input_b"][ii]]
    return red


def redefine_stack_index_map(tel_index, feedmask, prod, stack, reverse_stack):
    """Representative products made of present, unmasked inputs (``util/tools.py:359-414``)."""
    stack_new = stack.copy()
    stack_flag = np.zeros(stack_new.size, dtype=bool)
    for sind, (ii, jj) in enumerate(prod[stack["prod"]]):
        bi, bj = tel_index[ii], tel_index[jj]
        if (bi is None) or (bj is None) or not feedmask[bi, bj]:
            for ts in np.flatnonzero(reverse_stack["stack"] == sind):
                ti, tj = tel_index[prod[ts][0]], tel_index[prod[ts][1]]
                if (ti is not None) and (tj is not None) and feedmask[ti, tj]:
                    stack_new[sind]["prod"] = ts
                    stack_new[sind]["conjugate"] = reverse_stack[ts]["conjugate"]
                    stack_flag[sind] = True
                    break
        else:
            stack_flag[sind] = True
    return stack_new, stack_flag


def collate(vis, weight, input_flags, file_chan, file_freq, prod, tel_chan, tel_freq, feedmap, feedconj, weight_mode="natural",
            stack=None, reverse_stack=None, feedmask=None):
    """Returns ``(out_vis [nf_tel, npairs, nt] c64, out_weight f32, out_input_flags)``.

    ``stack`` / ``reverse_stack`` given: the second axis of ``vis`` is the file's stack axis (already stacked input).
    """
    input_ind = find_keys(list(tel_chan), list(file_chan), require_match=False)
    rev_input_ind = find_keys(list(file_chan), list(tel_chan), require_match=True)
    freq_ind = find_keys(list(file_freq), list(tel_freq), require_match=True)
    npairs = int(feedmap.max()) + 1
    nt = vis.shape[-1]
    spv = np.zeros((len(tel_freq), npairs, nt), dtype=np.complex64)
    spw = np.zeros((len(tel_freq), npairs, nt), dtype=np.float32)
    counter = np.zeros_like(spw)
    if stack is not None:
        stack_new, _ = redefine_stack_index_map(input_ind, feedmask, prod, stack, reverse_stack)
        ss_prod, ss_conj = prod[stack_new["prod"]], stack_new["conjugate"].astype(bool)
        stack_index = reverse_stack["stack"]
    else:
        ss_prod, ss_conj = prod, np.zeros(len(prod), dtype=bool)
        stack_index = np.arange(len(prod))
    if weight_mode != "inverse_variance":
        nprod_in_stack = calculate_redundancy(input_flags, prod, stack_index, vis.shape[1])
        if weight_mode == "uniform":
            nprod_in_stack = (nprod_in_stack > 0).astype(np.float32)
    for ss_pi, ((ii, ij), conj) in enumerate(zip(zip(ss_prod["input_a"], ss_prod["input_b"]), ss_conj)):
        bi, bj = input_ind[ii], input_ind[ij]
        if bi is None or bj is None:
            continue
        sp_pi = feedmap[bi, bj]
        if sp_pi < 0:
            continue
        if weight_mode == "inverse_variance":
            wss = weight[freq_ind, ss_pi]
        else:
            wss = (weight[freq_ind, ss_pi] > 0.0).astype(np.float32)
            wss[:] *= nprod_in_stack[np.newaxis, ss_pi, :]
        if bool(feedconj[bi, bj]) == bool(conj):  # transform.py:303 (conj is 0 for unstacked input)
            spv[:, sp_pi] += wss * vis[freq_ind, ss_pi]
        else:
            spv[:, sp_pi] += wss * vis[freq_ind, ss_pi].conj()
        spw[:, sp_pi] += wss**2 * invert_no_zero(weight[freq_ind, ss_pi])
        counter[:, sp_pi] += wss
    spv *= invert_no_zero(counter)
    spw = counter**2 * invert_no_zero(spw)
    return spv, spw, np.asarray(input_flags)[rev_input_ind, :]
