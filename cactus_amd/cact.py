"""Writer for CACT v1 tensor files holding already-quantized payloads.

The layout is the one the reference's converter writes (python/src/tensor_io.py:163-265,
save_tensor_with_header) and CactusGraph::mmap_weights reads (cactus/graph/graph_io.cpp:497-566):
an 80-byte header padded to the alignment, the fp16 group scales padded to the alignment, the payload.

    magic "CACT" | version u32 | flags u32 | alignment u32 | ndim u32 | dims 4 x u64 | precision u32 |
    data_bytes u64 | scales_bytes u64 | group_size u32 | num_groups u32

Precision values: 0 INT8, 1 FP16, 2 FP32, 3 INT4 (packed, two values per byte).  Quantizing is not done
here: the caller passes the bytes the file is to hold.
"""
import os
import struct

import numpy as np

MAGIC = b"CACT"
VERSION = 1
ALIGNMENT = 32
FLAG_HAS_SCALES = 1
INT8, FP16, FP32, INT4 = 0, 1, 2, 3


def _pad(n, align):
    return (-n) % align


def cact_bytes(shape, precision, data, scales=None, group_size=0, flags=None, alignment=ALIGNMENT, version=VERSION):
    """The bytes of one tensor file.  data / scales: arrays (any dtype; written as they lie in memory);
    scales, when given, are the fp16 group scales [rows, K / group_size]."""
    shape = [int(d) for d in shape]
    if not 1 <= len(shape) <= 4:
        raise ValueError("CACT tensors have 1 to 4 dimensions")
    payload = np.ascontiguousarray(data).tobytes()
    sc = b"" if scales is None else np.ascontiguousarray(scales).tobytes()
    num_groups = 0
    if scales is not None:
        if group_size <= 0 or shape[-1] % group_size:
            raise ValueError("the last dimension must be a multiple of the group size")
        num_groups = shape[-1] // group_size
    if flags is None:
        flags = FLAG_HAS_SCALES if scales is not None else 0
    head = MAGIC + struct.pack("<IIII", version, flags, alignment, len(shape))
    head += struct.pack("<4Q", *(shape + [0] * (4 - len(shape))))
    head += struct.pack("<IQQII", precision, len(payload), len(sc), group_size if scales is not None else 0, num_groups)
    assert len(head) == 80
    out = head + bytes(_pad(len(head), alignment)) + sc
    out += bytes(_pad(len(out), alignment))
    return out + payload


def write_cact(path, shape, precision, data, scales=None, group_size=0, **kw):
    with open(path, "wb") as f:
        f.write(cact_bytes(shape, precision, data, scales, group_size, **kw))


def write_qwen_folder(folder, weights, cfg):
    """A model folder in the reference's layout for the Qwen family (file names of
    models/model_qwen.cpp:17-44, config.txt keys of engine_model.cpp:403-539) from the host model dict of
    cactus_amd.steps: weights = {"emb": (B, Bs), "final_norm": v, "layers": [{"q": (B, Bs), ..., "in_norm": v, ...}]},
    matrices INT8 [N, K] (or packed INT4 uint8 [N, K/2]) with fp16 scales [N, K/128], vectors fp16.  The
    embedding doubles as the output projection (tie_word_embeddings)."""
    os.makedirs(folder, exist_ok=True)

    def mat(name, pair):
        B, Bs = pair
        int4 = B.dtype == np.uint8
        K = B.shape[1] * (2 if int4 else 1)
        write_cact(os.path.join(folder, name), (B.shape[0], K), INT4 if int4 else INT8, B,
                   np.asarray(Bs).view(np.uint16), 128)

    def vec(name, v):
        write_cact(os.path.join(folder, name), v.shape, FP16, np.asarray(v).view(np.uint16))

    mat("token_embeddings.weights", weights["emb"])
    vec("output_norm.weights", weights["final_norm"])
    names = {"q": "attn_q", "k": "attn_k", "v": "attn_v", "o": "attn_output", "gate": "ffn_gate", "up": "ffn_up",
             "down": "ffn_down", "in_norm": "input_norm", "post_norm": "post_attn_norm", "q_norm": "attn_q_norm",
             "k_norm": "attn_k_norm"}
    for i, L in enumerate(weights["layers"]):
        for key, val in L.items():
            fn = f"layer_{i}_{names[key]}.weights"
            if isinstance(val, tuple):
                mat(fn, val)
            else:
                vec(fn, val)
    with open(os.path.join(folder, "config.txt"), "w") as f:
        f.write(f"vocab_size={weights['emb'][0].shape[0]}\nhidden_dim={cfg['hidden']}\nnum_layers={cfg['layers']}\n"
                f"attention_heads={cfg['q_heads']}\nattention_kv_heads={cfg['kv_heads']}\n"
                f"attention_head_dim={cfg['head_dim']}\nffn_intermediate_dim={cfg['ffn']}\n"
                f"layer_norm_eps={cfg['eps']}\nrope_theta={cfg['theta']}\ntie_word_embeddings=true\n"
                "model_type=qwen\nprecision=INT8\nmodel_variant=default\n")
