// SceneDescription.h -- the parameter blocks the reference's scene items are constructed from, under
// the reference's names (src/Scene/SceneDescription.h:13-97) so that code written against it reads
// the same; no OptiX / Hypodermic types.
#pragma once

#include <array>
#include <cmath>
#include <string>
#include <utility>

namespace DeepestScatter
{
    using float3 = std::array<float, 3>;
    using Color = float3;
    using Meter = float;

    inline float3 normalize(const float3& v)
    {
        const float inv = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        return { v[0] * inv, v[1] * inv, v[2] * inv };
    }

    // How the estimator walks the cloud (SceneDescription.h:31-52).
    struct CloudRendering
    {
        using SampleStep = float;
        // :39-44 -- the three closest-hit programs of cloudRadianceMaterials.cu
        enum class Mode
        {
            SunAndSkyAllScatter,
            SunMultipleScatter,
            SunSingleScatter
        };

        CloudRendering(SampleStep step, Mode renderMode) : sampleStep(step), mode(renderMode) {}

        const SampleStep sampleStep;   // in box units; the reference uses 1/512 (installers.cpp:86)
        const Mode mode;
    };

    // What the cloud is (SceneDescription.h:54-82).
    struct CloudModel
    {
        using Size = Meter;
        using MeanFreePath = Meter;
        enum class Mipmaps : bool { Off = false, On = true };

        CloudModel(std::string path, Mipmaps mipmaps, Size sizeInMeters)
            : vdbPath(std::move(path)), mipmapsOn(mipmaps), size(sizeInMeters) {}

        const std::string vdbPath;                 // "procedural:<N>" or a .f32grid here (see Resources)
        const Mipmaps mipmapsOn;
        const Size size;                           // largest extent of the box in metres (7000 in main.cpp:63)
        const MeanFreePath meanFreePath = 10.0f;   // :80
    };

    struct Cloud
    {
        using Rendering = CloudRendering;
        using Model = CloudModel;

        Cloud(const Rendering& r, const Model& m) : rendering(r), model(m) {}

        const Rendering rendering;
        const Model model;
    };

    // :13-29.  The constructor normalises; libcloudtrace normalises once more, which is what
    // installSceneSetup followed by this constructor amounts to in the reference.
    struct DirectionalLight
    {
        DirectionalLight(const float3& towards, const Color& rgb, float power)
            : direction(normalize(towards)), color(rgb), intensity(power) {}

        const float3 direction;
        const Color color;
        const float intensity;
    };

    // :84-97
    struct SceneDescription
    {
        SceneDescription(const Cloud& c, const DirectionalLight& l) : cloud(c), light(l) {}

        const Cloud cloud;
        const DirectionalLight light;
    };
}
