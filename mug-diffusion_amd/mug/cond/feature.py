"""mug.cond.feature -- BeatmapFeatureEmbedder (mug/cond/feature.py:8-21)."""
import torch
import yaml

from mug.util import count_beatmap_features


class BeatmapFeatureEmbedder(torch.nn.Module):
    def __init__(self, path_to_yaml, embed_dim):
        super().__init__()
        with open(path_to_yaml) as f:
            self.feature_dicts = yaml.safe_load(f)
        self.embedding = torch.nn.Embedding(count_beatmap_features(self.feature_dicts), embed_dim)

    @torch.no_grad()
    def forward(self, x):
        """x (B, F) ids (any numeric dtype, like the reference's float tensor) -> (B, H, F)."""
        from mug._native import get_lib
        return get_lib().cond_embed(self.embedding.weight, x.long())

    def summary(self):
        pass
