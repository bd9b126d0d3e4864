"""Synthetic WikiWeb2M pages in the on-disk schema written by the reference's wikiweb2m/preprocess_data.py:116-145
(bytes columns, per-section arrays, flattened per-section image arrays) + a small offline tokenizer.
Used by the tests, the golden-vector generator and the trainer's `--dataset synthetic` smoke mode: there is no
network here, so neither the 2M-page dataset nor the HF tokenizers can be downloaded."""
import numpy as np
import pandas as pd

_WORDS = ("the of and in to a is was for on as by with that at from it an be this which or are his new first one has "
          "their were not but also its had who after two been other city river tower album band species frog election "
          "summary context page section image caption museum bridge mountain island village railway station church "
          "university football season company film music born died known american british french german national "
          "north south east west population area district county state world war century history house school").split()


def synthetic_pages(n_pages: int = 4, seed: int = 0, max_sections: int = 7, images_per_section: int = 2) -> pd.DataFrame:
    rng = np.random.RandomState(seed)

    def sent(lo, hi):
        return " ".join(rng.choice(_WORDS, size=rng.randint(lo, hi)).tolist())

    rows = []
    for p in range(n_pages):
        ns = int(rng.randint(2, max_sections + 1))
        parent = [-1] + [int(rng.randint(0, i)) for i in range(1, ns)]
        rows.append(dict(
            page_id=1000 + p,
            page_url=f"https://en.wikipedia.org/wiki/page_{p}".encode(),
            page_title=sent(1, 4).encode(),
            page_description=(sent(5, 20) + "\n" + sent(3, 9)).encode(),
            section_title=np.array([sent(1, 4).encode() for _ in range(ns)], dtype=object),
            section_depth=np.array([int(rng.randint(1, 4)) for _ in range(ns)]),
            section_heading=np.array([int(rng.randint(1, 4)) for _ in range(ns)]),
            section_parent_index=np.array(parent),
            section_summary=np.array([(sent(6, 14) + " .").encode() for _ in range(ns)], dtype=object),
            section_rest_sentence=np.array([(sent(10, 60) + "\n" + sent(4, 30)).encode() for _ in range(ns)], dtype=object),
            image_url=np.array([f"https://upload.example/{p}_{s}_{i}.jpg".encode() for s in range(ns) for i in range(images_per_section)], dtype=object),
            image_caption=np.array([sent(3, 9).encode() for s in range(ns) for i in range(images_per_section)], dtype=object),
        ))
    return pd.DataFrame(rows)


def synthetic_id_list(df: pd.DataFrame):
    return [(int(r.page_id), s) for r in df.itertuples() for s in range(len(r.section_title))]


def synthetic_tokenizer(extra_words=()):
    """WordLevel tokenizer with OPT-like specials: <pad>=1, </s>=2 used as BOS (prefix template) and EOS."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for w in list(_WORDS) + ["summarize", ":", ",", ".", "conext"] + list(extra_words):
        vocab.setdefault(w, len(vocab))
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Punctuation()])
    tok.post_processor = processors.TemplateProcessing(single="</s> $A", special_tokens=[("</s>", 2)])
    return PreTrainedTokenizerFast(tokenizer_object=tok, bos_token="</s>", eos_token="</s>", pad_token="<pad>", unk_token="<unk>")


def synthetic_images(df: pd.DataFrame, image_dir: str, seed: int = 0, keep: float = 0.55):
    """Image files for `synthetic_pages` in the layout the reference reads (`{page}_{section}_{idx}.{ext}`, ext taken from the image
    URL: wikiweb2m/data.py:132-136, preprocess_data.py:201-202).  A random subset of the (section, image) slots gets a small
    random RGB picture -- PNG bytes under the URL's `.jpg` name (PIL detects the format from the content; lossless, so the decoded
    pixels do not depend on the JPEG library) -- and every seventh written slot holds garbage instead, which exercises the reference's
    `except: continue` (the next image of that section is tried).  Returns {file name: "image" | "corrupt"}."""
    import os
    from PIL import Image
    rng = np.random.RandomState(seed)
    os.makedirs(image_dir, exist_ok=True)
    written = {}
    for r in df.itertuples():
        ns = len(r.section_title)
        urls = r.image_url.reshape(ns, -1)
        for s in range(ns):
            for i in range(urls.shape[1]):
                if rng.rand() >= keep:
                    continue
                ext = os.path.splitext(urls[s][i].decode())[1][1:]
                name = f"{int(r.page_id)}_{s}_{i}.{ext}"
                path = os.path.join(image_dir, name)
                if len(written) % 7 == 3:
                    with open(path, "wb") as f:
                        f.write(b"not an image")
                    written[name] = "corrupt"
                    continue
                h, w = int(rng.randint(8, 40)), int(rng.randint(8, 40))
                Image.fromarray(rng.randint(0, 256, size=(h, w, 3)).astype(np.uint8)).save(path, format="PNG")
                written[name] = "image"
    return written


def synthetic_pixel_values(img):
    """Offline stand-in for `AutoFeatureExtractor(...)(img.convert('RGB')).pixel_values[0]` (language_modelling/utils.py:21-23):
    RGB, nearest-neighbour resize to 224 x 224, [0, 1] floats, channels first -- deterministic, no pretrained preprocessor config."""
    import torch
    from PIL import Image
    a = np.asarray(img.convert("RGB").resize((224, 224), Image.NEAREST), dtype=np.float32) / 255.0
    return torch.from_numpy(a).permute(2, 0, 1).contiguous()
