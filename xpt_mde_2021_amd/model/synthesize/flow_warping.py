"""FlowWarpMultiScale with the reference's call signature (model/synthesize/flow_warping.py:11-71): the target view
reconstructed from every source frame by sampling it at (pixel grid - predicted optical flow), per flow scale."""
from ...utils import util_funcs as uf
from .bilinear_interp import BilinearInterpolation, FlowBilinearInterpolation


class FlowWarpMultiScale:
    def __call__(self, source_image, flow_ms):
        """
        :param source_image: source images [batch, numsrc, height, width, 3]
        :param flow_ms: predicted optical flow from source to target in multi scale,
                        list of [batch, numsrc, height/scale, width/scale, 2] (scale: 4, 8, 16, 32)
        :return: reconstructed target view in multi scale, list of [batch, numsrc, height/scale, width/scale, 3]
        """
        warped_targets = []
        for flow_sc in flow_ms:
            src_img_sc = self.reshape_source_images(source_image, flow_sc)
            pixel_coords_sc = self.flow_to_pixel_coordinates(flow_sc)
            warped_targets.append(BilinearInterpolation()(src_img_sc, pixel_coords_sc))
        return warped_targets

    def reshape_source_images(self, source_image, flow_sc):
        """flow_warping.py:35-50: TF2-bilinear resize of every source frame to the flow's resolution."""
        batch, numsrc, height_sc, width_sc, _ = flow_sc.shape
        batch, numsrc, height_ori, width_ori, _ = source_image.shape
        flat = source_image.reshape(batch * numsrc, height_ori, width_ori, 3)
        scaled = uf.resize_like_size(flat, height_sc, width_sc)
        return scaled.reshape(batch, numsrc, height_sc, width_sc, 3)

    def flow_to_pixel_coordinates(self, flow):
        """flow_warping.py:52-71: [batch, numsrc, h, w, 2(u,v)] -> [batch, numsrc, 2, h*w] = grid - flow."""
        return FlowBilinearInterpolation().flow_to_pixel_coordinates(flow)
