"""Group a rocprofv3 rocpd .db kernel trace into categories; print ms per category (total / per step)."""
import os, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
# steady-state window: between the last two marker kernels (torch.cuda._sleep -> "spin_kernel")
marks = [r[0] for r in db.execute("select start from kernels where name like '%spin_kernel%' order by start").fetchall()]
where = ""
if len(marks) >= 2:
    where = f" where start > {marks[-2]} and start < {marks[-1]} and name not like '%spin_kernel%'"
    span = (marks[-1] - marks[-2]) / 1e6
    print(f"window between markers: {span:.1f} ms wall, {span/steps:.2f} ms/step")
rows = db.execute(f"select name, count(*), sum(duration) from kernels{where} group by name").fetchall()
if "--top" in sys.argv:
    for name, n, dur in sorted(rows, key=lambda r: -r[2])[:int(os.environ.get("TOP", 25))]:
        print(f"  {dur/1e6/steps:8.3f} ms/step  {n/steps:7.1f} x  {name[:110]}")
cats = [("ours:conv2d (MFMA; 3x3 and 1x1: fwd, dgrad, wgrad; 7x7 stem + max-pool)", r"k_conv3x3|k_conv1x1|k_wgrad_sum|k_conv_f32|k_wino|k_stem_conv|k_maxpool3s2"),
        ("ours:head_tail", r"k_tail_|k_stats_|k_sum_slices|k_gtail"),
        ("ours:bn_act", r"k_bn_act|k_bn_bwd_reduce|k_bn_bwd_dx|k_bn_stats|k_bn_bwd_final"),
        ("ours:spconv", r"k_conv_mfma|k_conv_dma|k_wgrad_mfma|k_wgrad_bf16|k_wgrad_reduce|k_wgrad_generic|k_tile_masks|k_conv_generic|k_subm|k_down|k_set_bits|k_word|k_fill_perm|k_mark|k_emit|k_dense|k_offset_counts|k_row_masks"),
        ("ours:voxelize", r"k_insert|k_first|k_scan|k_assign|k_gather"),
        ("ours:lss/bev_pool", r"k_pool|k_bin|k_cell|k_fill\(|k_geometry|k_prepare|k_depth|k_transpose|k_lift|k_bwd|k_to_nhwc"),
        ("ours:distill", r"k_feat|k_rel|k_resp|k_mask|k_box"),
        ("miopen conv winograd", r"miopenSp3AsmConv"),
        ("miopen conv igemm fwd", r"igemm_fwd"), ("miopen conv igemm bwd", r"igemm_bwd"), ("miopen conv igemm wrw", r"igemm_wrw"),
        ("ck conv", r"kernel_grouped_conv|ck::"), ("miopen other conv/gemm", r"Cijk_|gemm|naive_conv|MIOpenConv|Im2|Col2"),
        ("miopen batchnorm", r"MIOpenBatchNorm"), ("miopen tensor ops", r"SubTensorOp|OpTensor|batched_transpose|transpose"),
        ("torch elementwise", r"elementwise_kernel|vectorized_elementwise|unrolled_elementwise"),
        ("torch reduce", r"reduce_kernel"), ("torch index/scatter/sort/topk", r"index|scatter|gather|sort|topk|cumsum|scan|rocprim|radix"),
        ("torch multi_tensor (optimizer/clip)", r"multi_tensor"), ("memcpy/fill", r"copyBuffer|fillBuffer|memset|Memcpy"),
        ("pooling/softmax/misc", r"pool|softmax|cat|Cat")]
agg = {}
for name, n, dur in rows:
    for c, pat in cats:
        if re.search(pat, name):
            break
    else:
        c = "other"
    a = agg.setdefault(c, [0, 0.0]); a[0] += n; a[1] += dur
tot = sum(v[1] for v in agg.values())
print(f"| category | launches/step | ms/step | % |\n|---|---|---|---|")
for c, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"| {c} | {n/steps:.0f} | {d/1e6/steps:.2f} | {100*d/tot:.1f} |")
print(f"| TOTAL | {sum(v[0] for v in agg.values())/steps:.0f} | {tot/1e6/steps:.2f} | 100 |")
