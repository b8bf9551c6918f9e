"""Drop-in for the reference's ``models.baseline_attention`` (AiR and OSIE variants).

  AiR :  baseline(embed_size=512, convLSTM_length=16, min_length=1, ratio=4, map_width=40, map_height=30)
         AiR/models/baseline_attention.py:187-189 ; forward(images, attention_maps, performances=None)
  OSIE:  baseline_osie(..., projected_label_length=18)  OSIE/models/baseline_attention.py:179-181 ; forward(images)
``baseline`` here is the AiR model (the north-star path); OSIE users import ``baseline_osie as baseline``.
Extra keyword (build-side extension): ``arch`` = "resnet50" (reference) | "resnet18".
"""
from .scanpath_model import ScanpathModel


class baseline(ScanpathModel):
    def __init__(self, embed_size=512, convLSTM_length=16, min_length=1, ratio=4, map_width=40, map_height=30,
                 arch="resnet50"):
        super().__init__("AiR", embed_size, convLSTM_length, min_length, ratio, map_width, map_height, arch)

    def forward(self, images, attention_maps, performances=None):
        return super().forward(images, attention_maps, performances)


class baseline_osie(ScanpathModel):
    def __init__(self, embed_size=512, convLSTM_length=16, min_length=1, ratio=4, map_width=40, map_height=30,
                 projected_label_length=18, arch="resnet50"):
        super().__init__("OSIE", embed_size, convLSTM_length, min_length, ratio, map_width, map_height, arch)

    def forward(self, images):
        return super().forward(images)
