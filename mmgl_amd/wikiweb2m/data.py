"""WikiWeb2M section-summarisation dataset: one training example = prompt/label ids + a fixed-shape bundle of
neighbor texts, neighbor images, position ids and interleave locations, so torch's default collate stacks it.

Mirrors reference wikiweb2m/data.py (`load_wikiweb2m`, `WikiWeb2M(args, df, id_list, tokenizer, visual_model_name)`,
the result-dict keys of `__getitem__` / `get_embedding_item`).  What differs, on purpose:
  * page lookup is an O(1) index built once (`page_id -> row`), not an O(N) dataframe scan per item (:316);
  * neighbor_mode "embedding" AND "cross_attention" both produce the neighbor bundle;
  * image files live under `image_dir` (constructor / env MMGL_IMAGE_DIR), not a hard-coded cluster path (:47);
  * position_type "laplacian" / "gnn": the reference calls utils.compute_LPE / utils.normalize_graph which do not exist
    (:434, :438); here they are defined (graph_pe.py: symmetric-normalised Laplacian eigenvectors, k = N-4 columns;
    D^-1/2 (A+I) D^-1/2) -- parity UNPINNED, only the consumer shapes are pinned (modelling_self_attention.py:137-139).
"""
import os
import pickle
from typing import Dict, List, Optional, Tuple

import pandas as pd
import torch
from PIL import Image

from . import graph_pe


def load_wikiweb2m(task, root="./wikiweb2m/raw"):
    """(train_df, val_df, test_df, id_list) from the parquet / pickle files written by preprocess_data.py (:13-31)."""
    dfs = [pd.read_parquet(os.path.join(root, f"wikiweb2m_{split}_large.parquet")) for split in ("train", "val", "test")]
    with open(os.path.join(root, f"{task}_id_split_large.pkl"), "rb") as f:
        id_list = pickle.load(f)
    return dfs[0], dfs[1], dfs[2], id_list


def _clean(text: str) -> str:
    return " ".join(text.replace("\n", " ").split())


class WikiWeb2M(torch.utils.data.Dataset):
    def __init__(self, args, df, id_list, tokenizer, visual_feature_extractor_model=None, image_dir=None):
        self.path = "./wikiweb2m/raw/"
        self.image_path = image_dir or os.environ.get("MMGL_IMAGE_DIR", "./wikiweb2m/raw/images")
        self.task = args.task
        self.context = args.context
        self.decoder_only = args.decoder_only
        self.neighbor_mode = args.neighbor_mode
        self.max_text_neighbors = args.max_text_neighbors
        self.max_image_neighbors = args.max_image_neighbors
        self.position_type = args.position_type
        self.df = df
        self._row_of = {int(pid): i for i, pid in enumerate(df["page_id"].tolist())}     # first occurrence wins, like .iloc[0]
        for i, pid in reversed(list(enumerate(df["page_id"].tolist()))):
            self._row_of[int(pid)] = i
        self.id_list = id_list
        self.tokenizer = tokenizer
        self.max_input_length = args.max_input_length
        self.max_output_length = args.max_output_length
        self.visual_feature_extractor = None
        if visual_feature_extractor_model is not None and self.context in ("section_all", "all"):
            if callable(visual_feature_extractor_model):
                self.visual_feature_extractor = visual_feature_extractor_model        # injected (tests / offline)
            else:
                from ..language_modelling import utils
                self.visual_feature_extractor = utils.get_feature_extractor_for_model(visual_feature_extractor_model)
        self.n_text_tokens = args.n_text_tokens
        self.n_visual_tokens = args.n_visual_tokens
        self.image_size = getattr(args, "image_size", 224)

    def __len__(self):
        return len(self.id_list)

    # ------------------------------------------------------------------------------------ raw text pieces
    def _page(self, page_id):
        return self.df.iloc[self._row_of[int(page_id)]]

    def get_page_info(self, d):
        """'<title>, <description>' with whitespace collapsed (:78-90)."""
        return _clean(", ".join([d["page_title"].decode(), d["page_description"].decode()]))

    def get_section_info(self, section_id, d, remove_summary=True):
        """remove_summary: (rest-of-section text, summary = the label); else 'summary, rest' as context text (:92-116)."""
        summary = d["section_summary"][section_id].decode()
        rest = d["section_rest_sentence"][section_id].decode()
        if remove_summary:
            return _clean(rest), _clean(summary)
        return _clean(", ".join([summary, rest]))

    def get_section_images(self, page_id, section_id, d):
        """First readable image of the section + its caption, else (None, None) (:118-144)."""
        section_num = d["section_title"].shape[0]
        image_urls = d["image_url"].reshape(section_num, -1)
        image_captions = d["image_caption"].reshape(section_num, -1)
        for image_id in range(image_urls[section_id].shape[0]):
            ext = os.path.splitext(image_urls[section_id][image_id].decode())[1][1:]
            file_name = os.path.join(self.image_path, f"{page_id}_{section_id}_{image_id}.{ext}")
            if not os.path.exists(file_name):
                continue
            try:
                img = Image.open(file_name)
                if self.visual_feature_extractor is None:
                    raise RuntimeError("no visual feature extractor")
                if callable(self.visual_feature_extractor) and not hasattr(self.visual_feature_extractor, "from_pretrained"):
                    pixels = self.visual_feature_extractor(img)
                else:
                    from ..language_modelling import utils
                    pixels = utils.get_pixel_values_for_model(self.visual_feature_extractor, img)
                return pixels, _clean(image_captions[section_id][image_id].decode())
            except Exception:
                continue
        return None, None

    # ------------------------------------------------------------------------------------ tokenisation helpers
    def _tok(self, text, max_length, padding):
        return self.tokenizer(text, max_length=max_length, padding=padding, truncation=True, return_tensors="pt")

    def _prompt_and_labels(self, input_ids_or_text, labels_text, prepadded=None):
        """Decoder-only: ids = prompt padded to max_input_length ++ ', summary: <label>' (BOS dropped, EOS appended) padded
        to max_output_length; labels = ids (pads are real tokens!).  Encoder-decoder: labels with pad -> -100 (:323-338)."""
        tk = self.tokenizer
        if prepadded is not None:
            model_inputs = prepadded
        else:
            model_inputs = tk.pad({"input_ids": [input_ids_or_text]}, max_length=self.max_input_length, padding="max_length", return_tensors="pt")
        if self.decoder_only:
            label_ids = self._tok(", summary: " + labels_text, self.max_output_length, "do_not_pad").input_ids[0]
            label_ids = torch.cat([label_ids[1:], torch.LongTensor([tk.eos_token_id])], dim=0)
            out = tk.pad({"input_ids": [label_ids]}, max_length=self.max_output_length, padding="max_length", return_tensors="pt")
            ids = torch.cat((model_inputs.input_ids[0], out.input_ids[0]), dim=0)
            return {"input_ids": ids, "attention_mask": torch.cat((model_inputs.attention_mask[0], out.attention_mask[0]), dim=0),
                    "labels": ids.clone()}
        labels = self._tok(labels_text, self.max_output_length, "max_length").input_ids[0]
        labels = torch.where(labels != 0, labels, torch.full_like(labels, -100))
        return {"input_ids": model_inputs.input_ids[0], "attention_mask": model_inputs.attention_mask[0], "labels": labels}

    def _blank_image(self):
        return torch.zeros((3, self.image_size, self.image_size))

    # ------------------------------------------------------------------------------------ raw mode (config 1 plumbing)
    def __getitem__(self, index):
        if self.neighbor_mode in ("embedding", "cross_attention"):
            return self.get_embedding_item(index)
        page_id, section_id = self.id_list[index]
        d = self._page(page_id)
        tk, nv = self.tokenizer, self.n_visual_tokens
        images, image_positions = [], []
        section_info, labels = self.get_section_info(section_id, d, remove_summary=True)

        def with_image(prefix_text, image, caption, budget):
            """text (+ ', conext: caption') truncated to `budget`, followed by n_visual_tokens placeholder ids (:183-199)."""
            if image is None:
                text, vis = prefix_text, torch.LongTensor(nv * [tk.pad_token_id])
                images.append(self._blank_image())
            else:
                text, vis = prefix_text + ", conext: " + caption, torch.LongTensor(nv * [-1])
                images.append(image)
            ids = self._tok(text, budget, "do_not_pad").input_ids[0]
            image_positions.append(ids.shape[0] + torch.arange(nv))
            return torch.cat([ids, vis], dim=0)

        if self.context == "section_only":
            input_ids = self._tok("summarize: " + section_info, self.max_input_length, "do_not_pad").input_ids[0]
        elif self.context == "section_all":
            image, caption = self.get_section_images(page_id, section_id, d)
            input_ids = with_image("summarize: " + section_info, image, caption, self.max_input_length - nv)
        elif self.context == "text_only":
            page_info = self.get_page_info(d)
            others = [self.get_section_info(c, d, remove_summary=False) for c in range(len(d["section_title"])) if c != section_id]
            text = "summarize: " + section_info + ", context: " + page_info + ", ".join(others)
            input_ids = self._tok(text, self.max_input_length, "do_not_pad").input_ids[0]
        elif self.context == "all":
            image, caption = self.get_section_images(page_id, section_id, d)
            input_ids = with_image("summarize: " + section_info, image, caption, self.max_input_length - nv)
            for c in range(len(d["section_title"])):
                if c == section_id:
                    continue
                ctx = self.get_section_info(c, d, remove_summary=False)
                cimg, ccap = self.get_section_images(page_id, c, d)
                text = ctx if cimg is None else ctx + ccap
                vis = torch.LongTensor(nv * [tk.pad_token_id if cimg is None else -1])
                ctx_ids = tk(text, padding="do_not_pad", truncation=False, return_tensors="pt").input_ids[0]
                if input_ids.shape[0] + ctx_ids.shape[0] + nv > self.max_input_length:
                    break
                images.append(self._blank_image() if cimg is None else cimg)
                image_positions.append(input_ids.shape[0] + ctx_ids.shape[0] + torch.arange(nv))
                input_ids = torch.cat([input_ids, ctx_ids, vis], dim=0)
            input_ids = input_ids[: self.max_input_length]
        else:
            raise ValueError(f"unknown context {self.context!r}")

        result = self._prompt_and_labels(input_ids, labels)
        if self.context in ("section_all", "all"):
            result["images"] = torch.stack(images, dim=0)
            result["image_positions"] = torch.cat(image_positions, dim=0)
        return result

    # ------------------------------------------------------------------------------------ the neighbor-batching collate
    def get_embedding_item(self, index):
        """Reference :296-469.  Neighbor slots, in order: page info; the section's own image then its caption; then for
        every other section its text, its image, its caption -- text slots capped at max_text_neighbors, image slots at
        max_image_neighbors.  `*_locations` = slot index in that interleaved order (padding slots last), position ids
        = 1-based order within the modality, 0 = padding."""
        page_id, section_id = self.id_list[index]
        d = self._page(page_id)
        section_info, labels = self.get_section_info(section_id, d, remove_summary=True)
        prompt = self._tok("summarize: " + section_info, self.max_input_length, "max_length")
        result = self._prompt_and_labels(None, labels, prepadded=prompt)

        texts: List[str] = []
        images: List[torch.Tensor] = []
        loc_text: List[int] = []
        loc_img: List[int] = []
        edges: List[Tuple[int, int]] = []
        graph_index = {section_id: 0}                   # node 0 = the input section; neighbors = slot + 1
        slot = 0

        def add_text(t):
            nonlocal slot
            texts.append(t)
            loc_text.append(slot)
            slot += 1
            return slot                                  # graph node id of the neighbor just added

        def add_image(im):
            nonlocal slot
            images.append(im)
            loc_img.append(slot)
            slot += 1
            return slot

        edges.append((0, add_text(self.get_page_info(d))))
        s_img, s_cap = self.get_section_images(page_id, section_id, d)
        if s_img is not None:
            img_node = add_image(s_img)
            edges.append((0, img_node))
            cap_node = add_text(s_cap)
            edges += [(0, cap_node), (img_node, cap_node)]

        prev_section = -1
        for c in range(len(d["section_title"])):
            if c == section_id:
                continue
            if len(texts) < self.max_text_neighbors:
                node = add_text(self.get_section_info(c, d, remove_summary=False))
                if prev_section > -1:
                    edges.append((prev_section, node))
                graph_index[c] = node
                prev_section = node
            if len(images) < self.max_image_neighbors:
                c_img, c_cap = self.get_section_images(page_id, c, d)
                if c_img is not None:
                    img_node = add_image(c_img)
                    edges.append((prev_section, img_node))
                    if len(texts) < self.max_text_neighbors:
                        cap_node = add_text(c_cap)
                        edges += [(prev_section, cap_node), (img_node, cap_node)]
        for c in range(len(d["section_parent_index"])):
            parent = d["section_parent_index"][c]
            if c in graph_index and parent in graph_index:
                edges.append((graph_index[c], graph_index[parent]))

        node_num = 1 + self.max_text_neighbors + self.max_image_neighbors
        edge_index = torch.LongTensor(edges).t().contiguous()
        if self.position_type == "laplacian":
            result["lpe"] = graph_pe.compute_LPE(edge_index, node_num)
        elif self.position_type == "gnn":
            result["graph"] = graph_pe.normalize_graph(graph_pe.dense_adjacency(edge_index, node_num))

        pos_text = list(range(1, len(texts) + 1))
        pos_img = list(range(1, len(images) + 1))
        while len(texts) < self.max_text_neighbors:
            texts.append("")
            pos_text.append(0)
            loc_text.append(slot)
            slot += 1
        while len(images) < self.max_image_neighbors:
            images.append(self._blank_image())
            pos_img.append(0)
            loc_img.append(slot)
            slot += 1

        nb = self._tok(texts, self.max_input_length, "max_length")
        result["neighbor_input_ids"] = nb.input_ids
        result["neighbor_attention_mask"] = nb.attention_mask
        result["neighbor_pos_ids"] = torch.LongTensor(pos_text)
        result["text_locations"] = torch.LongTensor(loc_text)
        result["neighbor_images"] = torch.stack(images, dim=0)
        result["neighbor_images_pos_ids"] = torch.LongTensor(pos_img)
        result["image_locations"] = torch.LongTensor(loc_img)
        return result
