"""Integer side of the LLaVA-Next image path, on the host (numpy): which ViT tokens make up an image's feature sequence and
where text tokens and image features land in the merged decoder input.  The device then only GATHERS rows
(vlr_gather_rows / vlr_merge_fwd) and scatters their gradients back - no S x H tensor is built on the host.

Mirrors, index for index,
  * transformers `select_best_resolution`, `image_size_to_num_patches`, `get_anyres_image_grid_shape`, `unpad_image` and
    `LlavaNextForConditionalGeneration.pack_image_features` ("spatial_unpad" + `image_newline`; call sites
    /root/reference/src/vlrlhf/models/LlavaNext/__init__.py:216-222, 255-259), and
  * `LlavaNextForRL._merge_input_ids_with_image_features` (same file :38-171).
"""
from typing import List, Sequence, Tuple

import numpy as np

SRC_ZERO = -(2 ** 31)         # csrc/dpo_ops.hip: merged position holds zeros
IGNORE_INDEX = -100


def select_best_resolution(original_size: Sequence[int], possible_resolutions) -> Tuple[int, int]:
    oh, ow = int(original_size[0]), int(original_size[1])
    best, max_eff, min_waste = None, 0, float("inf")
    for h, w in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            best, max_eff, min_waste = (int(h), int(w)), eff, waste
    return best


def image_size_to_num_patches(image_size, grid_pinpoints, patch_size: int) -> int:
    if not isinstance(grid_pinpoints, list):
        raise TypeError("grid_pinpoints should be a list of tuples or lists")
    h, w = select_best_resolution(image_size, grid_pinpoints)
    return len(range(0, h, patch_size)) * len(range(0, w, patch_size)) + 1


def pack_index(image_sizes, num_patches: Sequence[int], grid_pinpoints, image_size: int, patch_size: int):
    """-> (idx int32 [F], lens [n_img], newline_positions int32): packed feature row k of the batch is row idx[k] of the table
    [projector output of every evaluated tile, tile-major, g*g rows each | ONE image_newline row at index n_tiles*g*g]."""
    g = image_size // patch_size
    gg = g * g
    newline = int(sum(num_patches)) * gg
    out, lens = [], []
    base = 0
    for size, n in zip(image_sizes, num_patches):
        rows = [base * gg + i for i in range(gg)]                        # the base (whole image) tile first
        if n > 1:
            bh, bw = select_best_resolution(size, grid_pinpoints)
            nph, npw = bh // image_size, bw // image_size
            if nph * npw != n - 1:
                raise ValueError(f"image of size {tuple(size)}: {n - 1} tiles do not fill a {nph}x{npw} grid")
            ch, cw = nph * g, npw * g
            oh, ow = int(size[0]), int(size[1])
            y0, y1, x0, x1 = 0, ch, 0, cw
            if ow / oh > cw / ch:                                          # unpad_image: drop the padded rows / columns
                nh_ = int(round(oh * (cw / ow), 7))
                pad = (ch - nh_) // 2
                y0, y1 = pad, ch - pad
            else:
                nw_ = int(round(ow * (ch / oh), 7))
                pad = (cw - nw_) // 2
                x0, x1 = pad, cw - pad
            for y in range(y0, y1):
                ty, r = divmod(y, g)
                for x in range(x0, x1):
                    tx, c = divmod(x, g)
                    rows.append((base + 1 + ty * npw + tx) * gg + r * g + c)
                rows.append(newline)                                       # one image_newline per row of the un-padded grid
        else:
            rows.append(newline)
        out.append(rows)
        lens.append(len(rows))
        base += n
    idx = np.asarray([r for rows in out for r in rows], dtype=np.int32)
    return idx, np.asarray(lens, dtype=np.int64), np.nonzero(idx == newline)[0].astype(np.int32)


def merge_index(input_ids: np.ndarray, attention_mask: np.ndarray, labels, feature_lens: np.ndarray, image_token: int,
                padding_side: str = "left", dup: int = 1, ignore_index: int = IGNORE_INDEX):
    """feature_lens: one entry per `<image>` token of the (possibly duplicated) batch, in batch order; with dup > 1 the batch
    is `dup` identical halves sharing ONE table of F = sum(feature_lens) / dup packed rows.
    -> dict(src [Bn,S] int32 (>= 0 text token index | -(f+1) packed row | SRC_ZERO), mask, labels, pos, img_map, inv [dup,F], S)"""
    ids = np.asarray(input_ids)
    am = np.asarray(attention_mask)
    B, T = ids.shape
    fl = np.asarray(feature_lens, dtype=np.int64)
    lpad, rpad = bool((am[:, 0] == 0).any()), bool((am[:, -1] == 0).any())
    left = True
    if B > 1:
        if lpad and not rpad:
            left = True
        elif rpad and not lpad:
            left = False
        elif not lpad and not rpad:
            left = padding_side == "left"
        else:
            raise ValueError(f"both side of attention_mask has zero, invalid. {am}")
    is_img = ids == image_token
    n_img_row = is_img.sum(-1)
    if int(is_img.sum()) != fl.shape[0]:
        raise ValueError(f"Number of image tokens in input_ids ({int(is_img.sum())}) different from num_images ({fl.shape[0]}).")
    bounds = np.concatenate([[0], np.cumsum(n_img_row)])
    fl_row = np.asarray([int(fl[bounds[b]:bounds[b + 1]].sum()) for b in range(B)])
    seq_len = (am == 1).sum(-1) - n_img_row + fl_row
    S = int(seq_len.max())
    step = np.ones((B, T), dtype=np.int64)
    step[is_img] = fl
    new_pos = np.cumsum(step, -1) - 1
    if left:
        new_pos = new_pos + (S - 1 - new_pos[:, -1:])
    src = np.full((B, S), SRC_ZERO, dtype=np.int64)
    out_mask = np.zeros((B, S), dtype=np.int32)
    out_labels = np.full((B, S), ignore_index, dtype=np.int64)
    bi, ti = np.nonzero((~is_img) & (am == 1))
    dst = new_pos[bi, ti]
    src[bi, dst] = ti
    out_mask[bi, dst] = 1
    if labels is not None:
        out_labels[bi, dst] = np.asarray(labels)[bi, ti]
    free = np.ones((B, S), dtype=bool)
    free[bi, dst] = False
    idx = np.arange(S)[None]
    free &= ((S - idx) <= seq_len[:, None]) if left else (idx < seq_len[:, None])
    F_all = int(fl.sum())
    if int(free.sum()) != F_all:
        raise ValueError(f"image_to_overwrite.sum()={int(free.sum())} != num_image_features={F_all} The input provided to the model are "
                         "wrong. This prevents correct indexing and breaks batch generation.")
    if F_all % dup:
        raise ValueError("duplicated batch halves must carry the same images")
    F = F_all // dup
    k = np.arange(F_all)
    fb, fs = np.nonzero(free)                     # row-major: the k-th free slot takes packed row k (k mod F in its half)
    src[fb, fs] = -((k % F) + 1)
    inv = np.full((dup, F), -1, dtype=np.int32)
    inv[k // F, k % F] = (fb * S + fs).astype(np.int32)
    out_mask = out_mask | free.astype(np.int32)
    pos = np.cumsum(out_mask, -1) - 1
    pos[out_mask == 0] = 1
    return dict(src=src.astype(np.int32), mask=out_mask, labels=out_labels, pos=pos.astype(np.int32), img_map=free, inv=inv, S=S, F=F)
