"""GRSS2018 (reference loader/GRSS2018DataLoader.py:10-108): the hyperspectral raster has half the ground resolution
of the LiDAR raster; targets, scene shape and patches live on the LiDAR grid and every patch pixel takes the spectrum
of the HSI pixel that covers it (nearest, 2x).  The data set keeps both rasters at their own resolution -- in HBM too:
`hypel_gather_patches_2x_f32` does the 2x look-up while it cuts a batch of patches."""
import numpy

from hypelcnn_amd.common.common_nn_ops import BasicDataSet, shuffle_test_data_using_ratio, \
    shuffle_training_data_using_ratio, shuffle_training_data_using_size
from hypelcnn_amd.loader.DataLoader import DataLoader, SampleSet


class GRSS2018DataSet(BasicDataSet):
    casi_scale = 2  # LiDAR pixels per HSI pixel along each axis

    @staticmethod
    def _calculate_position(neighborhood, point, scale):
        actual_padding = int(neighborhood * scale)
        start_y = int(point[1] * scale) + neighborhood - actual_padding
        start_x = int(point[0] * scale) + neighborhood - actual_padding
        return start_x, start_y

    def get_data_point(self, point_x, point_y):
        """reference :30-44 with the per-pixel loop of :14-22 vectorised."""
        nb = self.neighborhood
        size = nb * 2 + 1
        start_x, start_y = self._calculate_position(nb, [point_x, point_y], 0.5)
        lidar_x, lidar_y = self._calculate_position(nb, [point_x, point_y], 1)
        half = (numpy.arange(size) * 0.5).astype(int)
        result = numpy.empty([size, size, self.casi.shape[2] + 1], dtype=self.casi.dtype)
        result[:, :, :-1] = self.casi[start_y + half[:, None], start_x + half[None, :], :]
        result[:, :, -1] = self.lidar[lidar_y:lidar_y + size, lidar_x:lidar_x + size, 0]
        return result


class GRSS2018DataLoader(DataLoader):
    """File-backed loader (needs the contest rasters as uncompressed TIFFs readable by common/tiff_io.py; the
    contest files themselves are not available in the build environment -- SyntheticDataLoader's "grss2018hr" preset
    produces the same two-resolution geometry)."""

    def __init__(self, base_dir):
        self.base_dir = base_dir

    def get_model_base_dir(self):
        return self.base_dir + "/2018/"

    def load_data(self, neighborhood, normalize):
        from hypelcnn_amd.common.tiff_io import imread
        casi = imread(self.get_model_base_dir() + "20170218_UH_CASI_S4_NAD83.tiff")[:, :, 0:-2]
        lidar = imread(self.get_model_base_dir() + "UH17c_GEF051.tif")[:, :, numpy.newaxis]
        lidar[numpy.where(lidar > 300)] = 0  # eliminate unacceptable values (:54)
        return GRSS2018DataSet(shadow_creator_dict=None, casi=casi, lidar=lidar, neighborhood=neighborhood,
                               normalize=normalize)

    def load_samples(self, train_data_ratio, test_data_ratio):
        """reference :66-88: classes 1..20 of the ground-truth window, shifted into scene coordinates."""
        from hypelcnn_amd.common.tiff_io import imread
        targets = imread(self.get_model_base_dir() + "2018_IEEE_GRSS_DFC_GT_TR.tif")
        y_delta, x_delta = 1202, 1194
        ys, xs = numpy.nonzero((targets >= 1) & (targets <= 20))
        order = numpy.lexsort((xs, ys, targets[ys, xs]))  # class-major, then row-major inside a class
        ys, xs = ys[order], xs[order]
        result = numpy.stack([xs.astype(int) + x_delta, ys.astype(int) + y_delta,
                              targets[ys, xs].astype(int) - 1], axis=1)
        if train_data_ratio < 1.0:
            train_set, validation_set = shuffle_training_data_using_ratio(result, train_data_ratio)
        else:
            train_set, validation_set = shuffle_training_data_using_size(self.get_class_count(), result,
                                                                         int(train_data_ratio), None)
        test_set, train_set = shuffle_test_data_using_ratio(train_set, test_data_ratio)
        return SampleSet(training_targets=train_set, test_targets=test_set, validation_targets=validation_set)

    def load_shadow_map(self, neighborhood, data_set):
        return None, None

    def get_class_count(self):
        return range(0, 20)

    def get_target_color_list(self):
        return None

    def get_band_measurements(self):
        return numpy.linspace(380, 1050, num=48)

    def get_samples_color_list(self):
        rng = numpy.random.RandomState(18)
        return rng.randint(0, 255, size=(20, 3)).astype(numpy.uint8)
