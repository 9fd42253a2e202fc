"""The one piece of platipy/imaging/generation the segmentation pipelines call: mask.extend_mask."""
