// SceneDescription.h -- the reference's scene PODs, same names and meaning
// (src/Scene/SceneDescription.h:13-97), minus the OptiX / Hypodermic types.
#pragma once

#include <array>
#include <cmath>
#include <string>

namespace DeepestScatter
{
    using Meter = float;
    using float3 = std::array<float, 3>;
    using Color = float3;

    inline float3 normalize(const float3& v)
    {
        const float inv = 1.0f / std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        return { v[0] * inv, v[1] * inv, v[2] * inv };
    }

    struct DirectionalLight
    {
        // SceneDescription.h:15-16: the constructor normalises (libcloudtrace normalises again, like
        // installSceneSetup + this constructor do in the reference)
        DirectionalLight(const float3& direction, const Color& color, float intensity)
            : direction(normalize(direction)), color(color), intensity(intensity) {}

        const float3 direction;
        const Color color;
        const float intensity;
    };

    struct Cloud
    {
        struct Rendering
        {
            using SampleStep = float;
            enum class Mode { SunAndSkyAllScatter, SunMultipleScatter, SunSingleScatter };   // :39-44

            Rendering(SampleStep sampleStep, Mode mode) : sampleStep(sampleStep), mode(mode) {}
            const SampleStep sampleStep;
            const Mode mode;
        };

        struct Model
        {
            using MeanFreePath = Meter;
            using Size = Meter;
            enum class Mipmaps : bool { Off = false, On = true };

            Model(const std::string& vdbPath, Mipmaps mipmaps, Size size)
                : vdbPath(vdbPath), mipmapsOn(mipmaps), size(size) {}

            const std::string vdbPath;
            const Mipmaps mipmapsOn;
            const Size size;
            const MeanFreePath meanFreePath = MeanFreePath{ Meter{ 10 } };               // :80
        };

        Cloud(const Rendering& rendering, const Model& model) : rendering(rendering), model(model) {}
        const Rendering rendering;
        const Model model;
    };

    struct SceneDescription
    {
        SceneDescription(const Cloud& cloud, const DirectionalLight& light) : cloud(cloud), light(light) {}
        const Cloud cloud;
        const DirectionalLight light;
    };
}
