"""BilinearInterpolation / FlowBilinearInterpolation with the reference's call signatures
(model/synthesize/bilinear_interp.py:5-32, 166-205), backed by the gfx950 sampler kernel (K3)."""
import torch

from ...hip import ops as _ops


class BilinearInterpolation:
    def __call__(self, image, pixel_coords, valid_mask=None):
        """
        :param image: source image [batch, numsrc, height, width, C]
        :param pixel_coords: (u, v[, 1]) float pixel coordinates [batch, numsrc, 2 or 3, height*width]
        :param valid_mask: zero pixels are INVALID in the result [batch, height, width, 1] or None
        :return: reconstructed image [batch, numsrc, height, width, C]; gradient flows to pixel_coords
        """
        return _ops.bilinear_sample(image, pixel_coords, valid_mask)


class FlowBilinearInterpolation:
    def __call__(self, image, flow):
        """image [batch*numsrc, height, width, C], flow [batch*numsrc, height, width, 2(u,v)]
        -> warped image [batch*numsrc, height, width, C]   (bilinear_interp.py:166-181)"""
        feature = image.unsqueeze(1)
        coords = self.flow_to_pixel_coordinates(flow.unsqueeze(1))
        return BilinearInterpolation()(feature, coords).squeeze(1)

    def flow_to_pixel_coordinates(self, flow):
        """flow [batch, numsrc, height, width, 2] -> coords [batch, numsrc, 2, height*width] = grid - flow
        (bilinear_interp.py:183-205)."""
        batch, numsrc, height, width, _ = flow.shape
        v, u = torch.meshgrid(torch.arange(height, dtype=flow.dtype, device=flow.device),
                              torch.arange(width, dtype=flow.dtype, device=flow.device), indexing="ij")
        uvgrid = torch.stack([u, v], dim=0).reshape(1, 1, 2, -1)
        uvflow = flow.reshape(batch, numsrc, -1, 2).permute(0, 1, 3, 2)
        return (uvgrid - uvflow).contiguous()
