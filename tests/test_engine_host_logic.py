"""Host-side logic of the engine that needs no GPU: how (branch, question) sequences are split into shared prompt
prefixes and compact own suffixes, the slot numbering of the two KV pools, and the work lists of the grouped
decode-attention pass."""
import torch

from llava_align_amd import ops
from llava_align_amd.engine import grouping_pays, IMAGE_TOKEN_INDEX, VddLlavaEngine, group_rows_by_prefix, preset

IMG = IMAGE_TOKEN_INDEX


class PlanOnly(VddLlavaEngine):
    """The planner only needs the config: skip weight / kernel set-up."""

    def __init__(self):
        self.cfg = preset("llava-1.5-7b")


def branches_for(ids_list, feats, use_unk=True, use_none=False):
    br = [("main", ids_list, feats)]
    if use_unk:
        br.append(("unk", [[0 if t == IMG else t for t in r] for r in ids_list], None))
    if use_none:
        br.append(("none", [[t for t in r if t != IMG] for r in ids_list], None))
    return br


def test_prefix_plan_pope_like():
    sys_tok = list(range(100, 135))                       # 35 system tokens
    img_a, img_b = torch.zeros(576, 8), torch.zeros(576, 8)
    ids = [sys_tok + [IMG] + [7, 8, 9], sys_tok + [IMG] + [7, 8, 9, 10], sys_tok + [IMG] + [5]]
    feats = [img_a, img_a, img_b]
    plan = PlanOnly()._plan(branches_for(ids, feats, use_none=True), 576, True)
    # prefixes: (sys + image a), (sys + image b), (sys + <unk>), (sys) -> 4 prefix slots, numbered in the prefix pool
    assert [p["T"] for p in plan["prefix"]] == [35 + 576, 35 + 576, 36, 35]
    assert [p["slot"] for p in plan["prefix"]] == [0, 1, 2, 3]
    suf = plan["suffix"]
    assert len(suf) == 9 and [s["slot"] for s in suf] == list(range(9))          # own pool: one compact slot per (branch, question)
    # main branch: questions 0, 1 share image a's prefix; question 2 has its own
    assert [s["pslot"] for s in suf[:3]] == [0, 0, 1] and all(s["plen"] == 611 and s["pos0"] == 611 for s in suf[:3])
    assert [s["T"] for s in suf[:3]] == [3, 4, 1]
    # <unk> branch: ONE token replaces the image slot and every question shares sys + <unk>
    assert all(s["pslot"] == 2 and s["plen"] == 36 for s in suf[3:6]) and suf[3]["tokens"] == [7, 8, 9]
    # image-token-dropped branch: shares the bare system prompt
    assert all(s["pslot"] == 3 and s["plen"] == 35 for s in suf[6:9])
    assert plan["prefill_tokens"] == (611 + 611 + 36 + 35) + 3 * (3 + 4 + 1)
    assert plan["unshared_tokens"] == (614 + 615 + 612) + (39 + 40 + 37) + (38 + 39 + 36)
    assert plan["max_len"] == 615


def test_plan_without_sharing_and_text_only_prompts():
    sys_tok = list(range(100, 110))
    img = torch.zeros(576, 8)
    ids = [sys_tok + [IMG] + [7, 8]]
    plan = PlanOnly()._plan(branches_for(ids, [img]), 576, False)
    assert plan["prefix"] == [] and [s["T"] for s in plan["suffix"]] == [10 + 576 + 2, 13] and all(s["plen"] == 0 for s in plan["suffix"])
    # a prompt with no image slot has nothing that marks a shareable prefix
    plan = PlanOnly()._plan([("main", [[1, 2, 3, 4]], None)], 576, True)
    assert plan["prefix"] == [] and plan["suffix"][0]["T"] == 4


def test_prompt_ending_in_the_unk_slot_keeps_one_token_for_the_logits():
    # main ids [5, 6, <image>]: in the <unk> branch the split point (after the <unk> token) is the END of the sequence;
    # the planner must leave the last token in the suffix, whose last-position logits the first decode step needs
    br = [("main", [[5, 6, IMG]], None), ("unk", [[5, 6, 0]], None)]
    plan = PlanOnly()._plan(br, 576, True)
    unk = plan["suffix"][1]
    assert unk["T"] == 1 and unk["tokens"] == [0] and unk["plen"] == 2 and unk["pos0"] == 2


def test_grouping_and_work_items():
    rows = [[0, 700, 0, 611], [1, 650, 0, 611], [2, 640, 1, 611], [3, 60, 2, 36], [4, 70, 2, 36], [5, 9, 0, 0]]
    groups, members = group_rows_by_prefix(rows)
    assert groups == [[0, 2, 0, 611], [2, 1, 1, 611], [3, 2, 2, 36]] and members == [0, 1, 2, 3, 4]      # row 5 has no prefix
    items = ops.prefix_work_items(groups)
    assert len(items) == 10 + 10 + 1 and items[0] == [0, 0, 0, 0] and sorted(items)[-1] == [2, 0, 0, 0]
    keys = [min(64, groups[g][3] - 64 * c) for g, _, c, _ in items]
    assert keys == sorted(keys, reverse=True) and keys[-3:] == [36, 35, 35]          # longest items first: the short ones fill the tail
    big = ops.prefix_work_items([[0, 40, 7, 36]])         # 40 rows -> three 16-row slices, one 64-key chunk each
    assert big == [[0, 0, 0, 0], [0, 16, 0, 0], [0, 32, 0, 0]]
    # several 64-key chunks per item: 611 keys -> 3 items of 256 keys (ragged last); item index, not chunk index, in column 2
    assert ops.prefix_work_items([[0, 6, 0, 611]], 4) == [[0, 0, 0, 0], [0, 0, 1, 0], [0, 0, 2, 0]]
    # one question in flight stays fully split; the bench shape (64 image groups of 6 + one 384-row group) walks a whole 611-key prefix per item
    assert ops.prefix_chunks_per_item([[0, 1, 0, 611], [1, 1, 1, 36]], 32) == 1
    bench_groups = [[6 * g, 6, g, 611] for g in range(64)] + [[384, 384, 64, 36]]
    assert ops.prefix_chunks_per_item(bench_groups, 32) == 10
    assert ops.prefix_chunks_per_item([[0, 16, 0, 4000]] * 512, 32) == 16        # capped


def test_grouping_only_when_it_removes_kv_traffic():
    # POPE batch: 6 questions per image share a 611-token prefix, every image-free row shares 36 tokens
    rows = [[i, 611 + 60, i // 6, 611] for i in range(384)] + [[384 + i, 36 + 60, 64, 36] for i in range(384)]
    groups, _ = group_rows_by_prefix(rows)
    assert grouping_pays(groups, rows)
    # LLaVA-Bench shape: one image per question -> only the short image-free prefixes are shared
    rows = [[i, 611 + 100, i, 611] for i in range(90)] + [[90 + i, 36 + 100, 90, 36] for i in range(90)] + [[180 + i, 35 + 100, 91, 35] for i in range(90)]
    groups, _ = group_rows_by_prefix(rows)
    assert len(groups) == 92 and not grouping_pays(groups, rows)
    assert not grouping_pays([], [])


def test_two_level_prefixes_prefill_the_system_prompt_once():
    """VERDICT r4 #2c: the 35 system-prompt tokens in front of every [sys + image] prefix are prefilled once; an image prefix runs its 576
    patch rows at positions 35.., attends [parent | own] and leaves rows 0..34 of its slot to a copy of the parent's K / V, so the slot
    the suffix pass and the decode steps read still holds 611 keys."""
    sys_tok = list(range(100, 135))
    img_a, img_b = torch.zeros(576, 8), torch.zeros(576, 8)
    ids = [sys_tok + [IMG] + [7, 8, 9], sys_tok + [IMG] + [7, 8, 9, 10], sys_tok + [IMG] + [5]]
    plan = PlanOnly()._plan(branches_for(ids, [img_a, img_a, img_b], use_none=True), 576, True)
    before = plan["prefill_tokens"]
    VddLlavaEngine._split_system_prompt(plan, min_saved=0)
    pre = plan["prefix"]
    assert [p["slot"] for p in pre] == [0, 1, 2, 3, 4] and pre[4]["tokens"] == sys_tok and pre[4]["img"] is None and pre[4]["T"] == 35
    for p in pre[:2]:                                                   # the two image prefixes: patch rows only, behind the parent
        assert p["T"] == 576 and p["full_T"] == 611 and p["pos0"] == 35 and p["cpos0"] == 35 and p["pslot"] == 4 and p["plen"] == 35 and p["tokens"] == []
    assert pre[2]["T"] == 36 and "cpos0" not in pre[2] and pre[3]["T"] == 35          # sys + <unk>, bare sys: untouched
    assert plan["prefill_tokens"] == before - 35                        # two images: one copy of the system prompt saved
    assert all(s["plen"] == 611 and s["pslot"] in (0, 1) for s in plan["suffix"][:3])   # the suffixes still see a 611-key prefix slot
    # a single image (or a short template) shares nothing worth a second pass
    plan1 = PlanOnly()._plan(branches_for(ids[:1], [img_a]), 576, True)
    n = len(plan1["prefix"])
    VddLlavaEngine._split_system_prompt(plan1)
    assert len(plan1["prefix"]) == n and "cpos0" not in plan1["prefix"][0]
    # ... and so do two images by default (one question + its VCD branch: a third prefill pass per layer to save 35 tokens is a loss;
    # the split has to save >= 128 prefill tokens = five images behind a 35-token template)
    for n_img, split in ((2, False), (4, False), (5, True)):
        imgs_ = [torch.zeros(576, 8) for _ in range(n_img)]
        plan_n = PlanOnly()._plan(branches_for([ids[0]] * n_img, imgs_), 576, True)
        VddLlavaEngine._split_system_prompt(plan_n)
        assert any("cpos0" in p_ for p_ in plan_n["prefix"]) == split, n_img
