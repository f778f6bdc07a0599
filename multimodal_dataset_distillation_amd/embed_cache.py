"""Frozen-text-encoder embedding caches and held-out tensors: the on-disk inputs either side of the
hot path (SURVEY 8b last row, 8f rank 4).

  {dataset}_{text_encoder}_text_embed.npz        key `bert_test_embed`  [n_test_captions, 768]
  {dataset}_{text_encoder}_train_text_embed.npz  key `bert_test_embed`  [n_train_pairs, 768]

are what reference utils.py:872-894 (`load_or_process_file`) reads, written by `textprocess` /
`textprocess_train` (reference distill.py:107-147).  Producing them needs BERT (outside the MI355X
path; unavailable offline); READING them needs nothing but numpy, and with them `text_syn` real
initialisation (reference distill.py:97-105, 228) and the nearest-caption decode (distill.py:89-95,
244) work end to end.  Files are loaded with allow_pickle=False.
"""
import os

import numpy as np
import torch

EMBED_KEY = "bert_test_embed"      # reference distill.py:121, 141; utils.py:885


def embed_cache_filename(dataset, text_encoder, file_type):
    """reference utils.py:885: f'{args.dataset}_{args.text_encoder}_{file_type}_embed.npz'"""
    return "%s_%s_%s_embed.npz" % (dataset, text_encoder, file_type)


def load_embed_cache(args, file_type, embed_dir=None):
    """reference load_or_process_file (utils.py:872-894) minus the 'process' branch: the cache must
    exist (creating it means running the frozen text encoder, which is not part of this path).
    Returns a float32 CPU tensor [n, d_txt]."""
    name = embed_cache_filename(args.dataset, args.text_encoder, file_type)
    d = embed_dir if embed_dir is not None else (getattr(args, "embed_dir", None) or ".")
    path = os.path.join(d, name)
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: the text-embedding cache is written by the reference's textprocess / "
            "textprocess_train (distill.py:107-147) with a frozen BERT; this engine only reads it" % path)
    print("Loading %s" % path)
    with np.load(path, allow_pickle=False) as z:
        if EMBED_KEY not in z.files:
            raise KeyError("%s has no '%s' array (keys: %s)" % (path, EMBED_KEY, z.files))
        arr = np.asarray(z[EMBED_KEY], dtype=np.float32)
    if arr.ndim != 2:
        raise ValueError("%s: expected a [n, d] array, got shape %s" % (path, arr.shape))
    return torch.from_numpy(arr)


def load_tensor_file(path, key=None):
    """A tensor saved as .pt (torch.save of a Tensor or a dict of Tensors; weights_only load), .npy or
    .npz.  `key` selects an entry of a dict / npz."""
    if path.endswith(".pt") or path.endswith(".pth"):
        obj = torch.load(path, map_location="cpu", weights_only=True)
        if isinstance(obj, dict):
            if key is None or key not in obj:
                raise KeyError("%s: need key %r, has %s" % (path, key, sorted(obj)))
            obj = obj[key]
        return torch.as_tensor(obj)
    if path.endswith(".npy"):
        return torch.from_numpy(np.load(path, allow_pickle=False))
    if path.endswith(".npz"):
        with np.load(path, allow_pickle=False) as z:
            if key is None or key not in z.files:
                raise KeyError("%s: need key %r, has %s" % (path, key, z.files))
            return torch.from_numpy(np.asarray(z[key]))
    raise ValueError("unsupported tensor file %s (.pt, .npy, .npz)" % path)


def get_images_texts(n, train_images, train_caption_embed, rng=None):
    """reference get_images_texts (distill.py:97-105): n random (image, caption) pairs of the training
    set; the caption side is the frozen text encoder's embedding of the pair's caption, which is row i
    of the train-caption cache (the reference's train datasets are lists of (image, caption) pairs and
    get_all_captions() walks them in the same order, data/flickr30k_dataset.py:67-83).
    train_images may be None (then only text_syn is drawn; image_syn comes back None).
    Uses np.random like the reference unless an np.random.Generator/RandomState is given."""
    m = train_caption_embed.shape[0]
    if train_images is not None and train_images.shape[0] != m:
        raise ValueError("train images (%d) and train caption embeddings (%d) are not aligned pairs"
                         % (train_images.shape[0], m))
    if n > m:
        raise ValueError("asked for %d pairs, the training set has %d" % (n, m))
    perm = (rng.permutation(m) if rng is not None else np.random.permutation(m))[:n]
    idx = torch.from_numpy(np.ascontiguousarray(perm)).long()
    image_syn = None if train_images is None else train_images[idx].float().contiguous()
    text_syn = train_caption_embed[idx].float().contiguous()
    return image_syn, text_syn, idx


def invert_txt2img(txt2img, n_img=None):
    """img2txt (list of caption-id lists, caption ids ascending) from txt2img -- how the reference's
    eval datasets build the pair of maps (data/flickr30k_dataset.py:105-117)."""
    t2i = np.asarray(txt2img, dtype=np.int64).reshape(-1)
    b = int(t2i.max()) + 1 if n_img is None else int(n_img)
    out = [[] for _ in range(b)]
    for t, i in enumerate(t2i.tolist()):
        out[i].append(t)
    return out


def load_eval_data(path, args=None):
    """Held-out retrieval set for evaluate_synset: an .npz with `images` [B,3,S,S] float32 (already
    normalised like the reference's test transform), `txt2img` [n_txt] and optionally
    `bert_test_embed` [n_txt, d] (otherwise the {dataset}_{text_encoder}_text_embed.npz cache)."""
    with np.load(path, allow_pickle=False) as z:
        images = torch.from_numpy(np.asarray(z["images"], dtype=np.float32))
        txt2img = np.asarray(z["txt2img"], dtype=np.int64)
        emb = torch.from_numpy(np.asarray(z[EMBED_KEY], dtype=np.float32)) if EMBED_KEY in z.files else None
    if emb is None:
        if args is None:
            raise KeyError("%s has no '%s' and no args to locate the cache" % (path, EMBED_KEY))
        emb = load_embed_cache(args, "text")
    if emb.shape[0] != txt2img.shape[0]:
        raise ValueError("caption embeddings (%d) and txt2img (%d) differ in length" % (emb.shape[0], txt2img.shape[0]))
    if txt2img.min() < 0 or txt2img.max() >= images.shape[0]:
        raise ValueError("txt2img refers to images outside [0, %d)" % images.shape[0])
    return images, emb, invert_txt2img(txt2img, images.shape[0]), txt2img.tolist()
