"""TEST INFRASTRUCTURE ONLY (see oracle/distill_ref.py header): numpy restatement of the reference's
retrieval metrics for the synthetic-set evaluation.

  itm_eval            <- reference epoch.py:219-244 (descending argsort per row, position of the best
                         ground-truth caption / of the ground-truth image, recall@1/5/10)
  similarity          <- reference epoch.py:106-107, :126, :148-149, :166-167: L2-normalised embeddings,
                         logit scale exp(log(1/0.07))
Pinned by tests/golden/itm_eval_small.npz, which oracle/gen_golden.py produces by executing the
reference's own `itm_eval` (AST-extracted from epoch.py; it is pure numpy) -- test_oracle.py.
"""
import numpy as np


def similarity(img_feat, txt_feat, scale=float(np.exp(np.log(1 / 0.07)))):
    a = img_feat / np.linalg.norm(img_feat, axis=1, keepdims=True)
    b = txt_feat / np.linalg.norm(txt_feat, axis=1, keepdims=True)
    return (scale * a.astype(np.float32) @ b.astype(np.float32).T).astype(np.float32)


def ranks(scores_i2t, scores_t2i, txt2img, img2txt):
    r_i = np.zeros(scores_i2t.shape[0], dtype=np.int32)
    for i, row in enumerate(scores_i2t):
        inds = np.argsort(row)[::-1]
        r_i[i] = int(np.min(np.where(np.isin(inds, img2txt[i]))[0]))
    r_t = np.zeros(scores_t2i.shape[0], dtype=np.int32)
    for j, row in enumerate(scores_t2i):
        inds = np.argsort(row)[::-1]
        r_t[j] = int(np.where(inds == txt2img[j])[0][0])
    return r_i, r_t


def recalls(r_i, r_t):
    tr = [100.0 * (r_i < k).sum() / len(r_i) for k in (1, 5, 10)]
    ir = [100.0 * (r_t < k).sum() / len(r_t) for k in (1, 5, 10)]
    return {"txt_r1": tr[0], "txt_r5": tr[1], "txt_r10": tr[2], "txt_r_mean": sum(tr) / 3,
            "img_r1": ir[0], "img_r5": ir[1], "img_r10": ir[2], "img_r_mean": sum(ir) / 3,
            "r_mean": (sum(tr) + sum(ir)) / 6}


def itm_eval(scores_i2t, scores_t2i, txt2img, img2txt):
    return recalls(*ranks(scores_i2t, scores_t2i, txt2img, img2txt))


def nearest_neighbor(sentences, query_embeddings, database_embeddings):
    """reference distill.py:89-95, with sklearn's cosine_similarity written out
    (x.y / (|x||y|), sklearn.metrics.pairwise.cosine_similarity) and np.argmax (first maximum)."""
    q = np.asarray(query_embeddings, dtype=np.float64)
    b = np.asarray(database_embeddings, dtype=np.float64)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    bn = b / np.linalg.norm(b, axis=1, keepdims=True)
    out = []
    for row in qn:
        out.append(sentences[int(np.argmax(bn @ row))])
    return out
